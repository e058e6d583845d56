/* Device code of the SPH pair path for gfx950: cell hash, LDS-staged radix sort, cell table,
 * candidate lists, the fused pair/update kernel and the small reductions around it.
 * Included once, by sph_amd.hip. */
#ifndef SPH_KERNELS_H
#define SPH_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sph_types.h"

#define SPH_BLOCK 256
#define SPH_HALO_ID 0xFFFFFFFFu /* id[] of a particle owned by a neighbouring slab */

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */

/* Squared distance with the exact operation order of the reference's fixed-radius search
 * (nanoflann L2 adaptor: ((dx*dx)+dy*dy)+dz*dz, no fused multiply-add).  Membership of the
 * neighbour set is decided on this value and must not depend on contraction. */
__device__ __forceinline__ double dist2_exact(double dx, double dy, double dz) {
#pragma clang fp contract(off)
    double r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

__device__ __forceinline__ int cell_coord(double x, double lo, double inv, int n) {
    double c = floor((x - lo) * inv);
    c = fmax(c, 0.0);               /* NaN -> 0 */
    c = fmin(c, (double)(n - 1));
    return (int)c;
}

/* Index of a column of cells (see GridParams): 3-D columns are grouped into bricks of 2^ts[0] x
 * 2^ts[1], bricks numbered with x slowest; in 2-D the column is the x cell.  The sort key of a
 * particle is column * nrun + its fine cell along the run axis. */
template <int DIM>
__device__ __forceinline__ uint32_t column_index(const GridParams& g, int cx, int cy) {
    if (DIM == 2) return (uint32_t)cx;
    const uint32_t ux = (uint32_t)cx, uy = (uint32_t)cy;
    const uint32_t s0 = g.ts[0], s1 = g.ts[1];
    const uint32_t brick = (ux >> s0) * (uint32_t)g.nt[1] + (uy >> s1);
    const uint32_t local = ((ux & ((1u << s0) - 1u)) << s1) | (uy & ((1u << s1) - 1u));
    return (brick << (s0 + s1)) | local;
}

/* Workgroups are dealt round-robin to the eight XCDs, each with its own L2.  This maps
 * blockIdx.x (grid a multiple of 8) so that an XCD works through chunks of XCD_CHUNK consecutive
 * blocks of the cell-ordered particles: what a workgroup gathers is then mostly what its
 * neighbours on the same L2 have just fetched, and the chunks are dealt to the XCDs in turn, so
 * that regions with little work per particle (dry floor, walls) are shared out evenly.
 * (Placement only changes speed, never results.) */
#ifndef XCD_CHUNK
#define XCD_CHUNK 64u
#endif
__device__ __forceinline__ uint32_t xcd_contiguous_block() {
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t x = blockIdx.x & 7u, l = blockIdx.x >> 3;
    const uint32_t row = l / XCD_CHUNK, r = l % XCD_CHUNK;
    /* rows of 8 chunks; the last row's chunks are shorter */
    const uint32_t full = per_xcd / XCD_CHUNK;
    const uint32_t width = row < full ? XCD_CHUNK : per_xcd - full * XCD_CHUNK;
    return row * 8u * XCD_CHUNK + x * width + r;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

/* data-parallel-primitive moves inside a quad of lanes: CTRL is a quad_perm pattern */
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double quad_perm_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = quad_perm_i<CTRL>((int)(b & 0xFFFFFFFFll));
    const int hi = quad_perm_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define QP_XOR1 0xB1 /* lanes [1,0,3,2] */
#define QP_XOR2 0x4E /* lanes [2,3,0,1] */
__device__ __forceinline__ double quad_sum(double v) {
    v += quad_perm_d<QP_XOR1>(v);
    v += quad_perm_d<QP_XOR2>(v);
    return v;
}
__device__ __forceinline__ double quad_max(double v) {
    v = fmax(v, quad_perm_d<QP_XOR1>(v));
    v = fmax(v, quad_perm_d<QP_XOR2>(v));
    return v;
}

/* x and (x moved by W lanes to its partner block), W = 16 or 32, for a double: the row-swap
 * instructions exchange the odd blocks of one copy with the even blocks of the other (vector ALU,
 * no trip through the LDS crossbar), after which the two copies hold, lane by lane, the two
 * numbers a shuffle by W would pair. */
template <int W>
__device__ __forceinline__ void rows_pair(double x, double& a, double& b) {
    const uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x);
    if (W == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        a = __hiloint2double((int)h[0], (int)l[0]);
        b = __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        a = __hiloint2double((int)h[0], (int)l[0]);
        b = __hiloint2double((int)h[1], (int)l[1]);
    }
}
template <int W>
__device__ __forceinline__ double rows_pair_sum(double x) {
    double a, b;
    rows_pair<W>(x, a, b);
    return a + b;
}
template <int W>
__device__ __forceinline__ double rows_pair_max(double x) {
    double a, b;
    rows_pair<W>(x, a, b);
    return fmax(a, b);
}
/* Wave-wide sum and maximum of a value that is the same in the four lanes of every quad, counted
 * once per quad, in every lane, without the LDS crossbar: mirror images within eight and within
 * sixteen lanes pair whole quads, the row swaps pair rows.  The order of additions is fixed. */
__device__ __forceinline__ double quads_allsum(double v) {
    v += quad_perm_d<0x141>(v); /* row_half_mirror */
    v += quad_perm_d<0x140>(v); /* row_mirror */
    v = rows_pair_sum<16>(v);
    return rows_pair_sum<32>(v);
}
__device__ __forceinline__ double quads_allmax(double v) {
    v = fmax(v, quad_perm_d<0x141>(v));
    v = fmax(v, quad_perm_d<0x140>(v));
    v = rows_pair_max<16>(v);
    return rows_pair_max<32>(v);
}
/* The same for a value of its own in every lane: partners 1 and 2 apart in the quad first (after
 * the quad steps every lane of a quad holds the quad's total, so the mirrors pair whole quads). */
__device__ __forceinline__ double wave_allsum(double v) {
    v += quad_perm_d<QP_XOR1>(v);
    v += quad_perm_d<QP_XOR2>(v);
    return quads_allsum(v);
}
__device__ __forceinline__ double wave_allmax(double v) {
    v = fmax(v, quad_perm_d<QP_XOR1>(v));
    v = fmax(v, quad_perm_d<QP_XOR2>(v));
    return quads_allmax(v);
}
__device__ __forceinline__ uint32_t quads_allmax(uint32_t v) {
    v = max(v, (uint32_t)quad_perm_i<0x141>((int)v));
    v = max(v, (uint32_t)quad_perm_i<0x140>((int)v));
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max((uint32_t)a[0], (uint32_t)a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((uint32_t)b[0], (uint32_t)b[1]);
}

/* ------------------------------------------------------------------------------------------ */
/* bounding box of finite positions (per-block partials; the host finishes)                   */

__global__ __launch_bounds__(SPH_BLOCK) void k_bbox(const Rec* __restrict__ rec, uint32_t n, int dim,
                                                    const uint32_t* __restrict__ drop /* may be null */,
                                                    double* __restrict__ part /* [grid][6] */) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x; i < n; i += gridDim.x * SPH_BLOCK) {
        bool ok = !(drop && drop[i]);
        for (int d = 0; d < dim; ++d) ok = ok && isfinite(rec[i].xv[d][0]);
        if (ok)
            for (int d = 0; d < dim; ++d) {
                lo[d] = fmin(lo[d], rec[i].xv[d][0]);
                hi[d] = fmax(hi[d], rec[i].xv[d][0]);
            }
    }
    __shared__ double s[SPH_BLOCK / 64][6];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int d = 0; d < 3; ++d) {
        lo[d] = wave_min(lo[d]);
        hi[d] = wave_max(hi[d]);
    }
    if (l == 0)
        for (int d = 0; d < 3; ++d) { s[w][d] = lo[d]; s[w][3 + d] = hi[d]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = s[0][threadIdx.x];
        for (int k = 1; k < SPH_BLOCK / 64; ++k)
            v = threadIdx.x < 3 ? fmin(v, s[k][threadIdx.x]) : fmax(v, s[k][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cell hash                                                                                  */

template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_hash(const Rec* __restrict__ rec, uint32_t n, GridParams g,
                                                    const uint32_t* __restrict__ drop /* may be null */,
                                                    uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (drop && drop[i]) { /* handed to a neighbouring slab: sorts behind every cell */
        keys[i] = g.ncells;
        return;
    }
    const int cx = cell_coord(rec[i].xv[0][0], g.lo[0], g.inv[0], g.n[0]);
    const int cy = DIM == 3 ? cell_coord(rec[i].xv[1][0], g.lo[1], g.inv[1], g.n[1]) : 0;
    const int cf = cell_coord(rec[i].xv[DIM - 1][0], g.lo[DIM - 1], g.invf, (int)g.nrun);
    keys[i] = column_index<DIM>(g, cx, cy) * g.nrun + (uint32_t)cf;
}

/* ------------------------------------------------------------------------------------------ */
/* exclusive scan of uint32 (three phases)                                                    */

#define SCAN_ITEMS 8
#define SCAN_TILE (SPH_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[SPH_BLOCK / 64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (l >= o) inc += t;
    }
    if (l == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SPH_BLOCK / 64; ++k) {
        if (k < w) base += wsum[k];
        tot += wsum[k];
    }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SPH_BLOCK) void k_scan_reduce(const uint32_t* __restrict__ in, uint32_t n,
                                                           uint32_t* __restrict__ bsum) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

/* one block; turns block sums into exclusive offsets */
__global__ __launch_bounds__(SPH_BLOCK) void k_scan_bsums(uint32_t* __restrict__ bsum, uint32_t nb) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += SPH_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(SPH_BLOCK) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          uint32_t n, const uint32_t* __restrict__ bsum) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t run = block_exclusive_scan(s, nullptr) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* stable LSD radix sort of (cell key, particle index), 8 bits per pass, LDS-staged           */

#define RS_ITEMS 16
#define RS_WAVES (SPH_BLOCK / 64)
#define RS_WCHUNK (64 * RS_ITEMS)
#define RS_TILE (SPH_BLOCK * RS_ITEMS)

/* hist[digit * nblocks + block] = number of keys of this block's tile with that digit */
__global__ __launch_bounds__(SPH_BLOCK) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                       uint32_t* __restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t idx = base + r * SPH_BLOCK + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

/* Scatter pass.  Tile order is (wave, round, lane) = ascending index, each wave owning a
 * contiguous chunk; ranks are counted in that order, so equal digits keep their order.
 * Keys and values are first permuted inside LDS so that the global writes of one digit
 * are contiguous. vals_in == nullptr means "identity". */
__global__ __launch_bounds__(SPH_BLOCK) void k_rs_scatter(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                          const uint32_t* __restrict__ hist_scanned,
                                                          uint32_t nblocks) {
    __shared__ uint32_t wcnt[RS_WAVES][256];
    __shared__ uint32_t dstart[256];
    __shared__ uint32_t gofs[256];
    __shared__ uint32_t skey[RS_TILE];
    __shared__ uint32_t sval[RS_TILE];
    __shared__ uint32_t wtot[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < RS_WAVES * 256; k += SPH_BLOCK) (&wcnt[0][0])[k] = 0;
    __syncthreads();

    const uint32_t tile_base = blockIdx.x * RS_TILE;
    const uint32_t wbase = tile_base + wave * RS_WCHUNK;
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
    volatile uint32_t* wc = wcnt[wave];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t idx = wbase + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
        const uint32_t digit = valid ? ((key[r] >> shift) & 255u) : 255u;
        unsigned long long mask = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t lane_rank = __popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t group = __popcll(mask);
        const uint32_t prev = wc[digit];
        rank[r] = prev + lane_rank;
        __builtin_amdgcn_wave_barrier();
        if (lane_rank == 0) wc[digit] = prev + group;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* thread = digit: prefix over waves, then scan over digits */
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
        /* exclusive scan of `run` over the 256 digits */
        uint32_t inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        const uint32_t ex = base + inc - run;
        dstart[tid] = ex;
        gofs[tid] = hist_scanned[tid * nblocks + blockIdx.x] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t idx = wbase + r * 64 + lane;
        const uint32_t digit = idx < n ? ((key[r] >> shift) & 255u) : 255u;
        const uint32_t pos = dstart[digit] + wcnt[wave][digit] + rank[r];
        skey[pos] = key[r];
        sval[pos] = val[r];
    }
    __syncthreads();
    const uint32_t nvalid = n - tile_base < RS_TILE ? n - tile_base : RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const uint32_t p = r * SPH_BLOCK + tid;
        if (p < nvalid) {
            const uint32_t kk = skey[p];
            const uint32_t out = gofs[(kk >> shift) & 255u] + p;
            keys_out[out] = kk;
            vals_out[out] = sval[p];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* physical reorder into cell order                                                           */

struct ReorderArgs {
    const uint32_t* perm; /* new position -> old position */
    uint32_t n;
    int nd, nf, nr;
    const double* sd[18];
    double* dd[18];
    const float* sf[6];
    float* df[6];
    const Rec* rs[2];
    Rec* rd[2];
    const uint32_t* ids;
    uint32_t* idd;
    const uint32_t* gids; /* global ids (slab runs), may be null */
    uint32_t* gidd;
};

__global__ __launch_bounds__(SPH_BLOCK) void k_reorder(ReorderArgs a) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t s = a.perm[i];
    for (int k = 0; k < a.nd; ++k) a.dd[k][i] = a.sd[k][s];
    for (int k = 0; k < a.nf; ++k) a.df[k][i] = a.sf[k][s];
    for (int k = 0; k < a.nr; ++k) a.rd[k][i] = a.rs[k][s];
    a.idd[i] = a.ids[s];
    if (a.gids) a.gidd[i] = a.gids[s];
}

/* ------------------------------------------------------------------------------------------ */
/* cell table: cstart[c] = first sorted position whose key is >= c, for c = 0 .. ncells         */
/* (ncells + 1 entries; the particles of cells a..b are [cstart[a], cstart[b + 1]))            */

__global__ __launch_bounds__(SPH_BLOCK) void k_cell_start(const uint32_t* __restrict__ keys, uint32_t n,
                                                          uint32_t index_base, uint32_t ncells,
                                                          uint32_t* __restrict__ cstart) {
    const uint32_t c = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (c > ncells) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    cstart[c] = index_base + lo;
}

/* ------------------------------------------------------------------------------------------ */
/* Wall particles at rest.  Inside a Newmark-beta step the wall particles' f and V are reset    */
/* after every Forces call (WCXSPH3D.cpp:260-264), so of everything Forces computes for a wall   */
/* particle only Rrho and mu survive.  A pair of two wall particles adds -m*(Vij.Grad) to Rrho   */
/* and H*(Vij.Rij)/(...) to the mu candidates: both exactly zero while every wall particle has   */
/* v = 0, and with v = 0, V = 0 and f = 0 in state n the update leaves wall positions and        */
/* velocities unchanged.  While that holds, wall-wall pairs can be left out of the candidate     */
/* lists without changing any stored value.  flag[0] is raised if it does not hold.              */
struct WallRestArgs {
    uint32_t n, wall_below;
    int dim, xsph;
    const Rec* rec;
    const double* nv[3];
    const double* nV[3];
    const double* nf[3];
    const uint32_t* id;
};
__global__ __launch_bounds__(SPH_BLOCK) void k_walls_at_rest(WallRestArgs a, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= a.n || a.id[i] >= a.wall_below) return;
    bool rest = true;
    for (int d = 0; d < a.dim; ++d) {
        rest = rest && a.rec[i].xv[d][1] == 0.0 && a.nv[d][i] == 0.0 && a.nf[d][i] == 0.0;
        if (a.xsph) rest = rest && a.nV[d][i] == 0.0;
    }
    if (!rest) flag[0] = 1u;
}

/* ------------------------------------------------------------------------------------------ */
/* candidate lists: every j != i with |xi-xj|^2 <= rl2 at build time (and the few the lattice */
/* test lets in just beyond), in cell-range order                                            */
/* then ascending j.  One row of `stride` entries per particle: entry k of particle i sits at  */
/* nbr[i*stride + k], so the rounds of the pair loop walk along cache lines they have already  */
/* touched.  The stride is a multiple of 4 (every 16-byte group is aligned) and, whenever the  */
/* library chooses it, of 32: rows then start on 128-byte lines, and no 64-byte group of a row */
/* straddles two lines.                                                                        */

/* ------------------------------------------------------------------------------------------ */
/* Step lists.  The candidate lists carry a generous skin so that they survive several steps;  */
/* a sweep only needs the candidates within the search radius plus what the particles move     */
/* until the lists are drawn again.  Once per step (after the first sweep, on the positions the  */
/* other seven start from, which differ from each other by dt^2-sized corrections) and after     */
/* every rebuild, k_draw_lists copies out of every candidate row the entries within               */
/* r_in = R*(1+margin), in the row's order; the pair kernel walks those.  Membership is still      */
/* decided exactly by the pair kernel: the draw only has to keep every pair within r_in, so it      */
/* works on 8-byte entries: the low 16 bits of the particle's coordinates on a lattice of spacing    */
/* q = r_list / 8192 (x and y in one word, z in the low half of the other).  The two particles of a    */
/* candidate pair are less than 2 r_list apart along any axis, far inside the +-4 r_list that a 16-bit  */
/* difference spans, so the wrapped differences are the true lattice differences, whatever the size of  */
/* the domain; rounding to the lattice moves a particle by at most q/2 along each axis, so the distance */
/* of two entries is off by less than two spacings, by which the radius is widened.  All in integers.   */
/* The list build tests its candidates on the same entries (there bit 31 of the second word marks a     */
/* wall particle): it meets particles up to two cell edges away, whose wrapped differences are still    */
/* exact while a cell is no wider than 2 r_list, and beyond that can only make a far particle look      */
/* near -- a row then holds an entry too many, never one too few.                                       */
#define DRAW_LATTICE_PER_RLIST 8192.0
#define QPOS_WALL 0x80000000u
struct QLattice {
    double org[3]; /* lattice origin (the low corner of the cell box: keeps the coordinates small) */
    double invq;   /* 1 / spacing */
};
typedef short s16x2_t __attribute__((ext_vector_type(2)));
/* the 8-byte lattice entry of a position (no wall mark) */
__device__ __forceinline__ uint2 lattice_entry(const QLattice& L, int dim, double x, double y, double z) {
    auto low16 = [&](double v, double org) -> uint32_t {
        const double r = rint((v - org) * L.invq);
        return (uint32_t)(r - 65536.0 * floor(r * (1.0 / 65536.0))) & 0xFFFFu; /* exact: r is an integer */
    };
    uint2 p;
    p.x = low16(x, L.org[0]) | (low16(y, L.org[1]) << 16);
    p.y = dim == 3 ? low16(z, L.org[2]) : 0u;
    return p;
}
/* id != nullptr: particles whose id is below bound_parts are marked as wall particles */
__global__ __launch_bounds__(SPH_BLOCK) void k_fill_qpos(const Rec* __restrict__ rec, uint32_t m, QLattice L, int dim,
                                                         const uint32_t* __restrict__ id, uint32_t bound_parts,
                                                         uint2* __restrict__ qpos) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= m) return;
    uint2 p = lattice_entry(L, dim, rec[i].xv[0][0], rec[i].xv[1][0], rec[i].xv[2][0]);
    if (id && id[i] < bound_parts) p.y |= QPOS_WALL;
    qpos[i] = p;
}

/* Four lanes per particle.  The first version of this kernel (quad-wide exchange of the row
 * entries, 16-byte single-precision positions, a ballot and a scattered 4-byte store per kept
 * entry) took 2.9 ms at 12.7 M particles, and taking it apart showed the parts adding up instead of
 * overlapping: 1.1 ms of arithmetic alone, 0.2 ms for the row reads, 0.8 ms for the gathers
 * (whatever their size, their addresses or the number in flight), 0.8 ms for the stores.  So the
 * work per entry is kept small: a lane reads 16 bytes of the candidate row and deals with those
 * four entries itself (no exchange inside the quad); the distance test is two packed 16-bit
 * subtractions, two integer dot products and a compare; the keep flags of sixteen entries are
 * combined across the quad into one mask in the row's order, from which every lane counts the
 * keepers in front of its own; and the keepers are collected in LDS and written out as whole
 * 16-byte groups at the end (rows with more than DRAW_STAGE keepers write the rest straight to
 * memory): 1.98 ms, 0.18 ms of it k_fill_qpos.  Slots of the last group past the count hold the
 * particle's own index: the pair kernel fetches what a row names before it looks at the count. */
#define DRAW_GROUPS 2        /* 16-entry groups of the row per step (12.7 M particles, 61 candidates each: 1.98 ms with 1 or 2,
                              * 2.27 ms with 4; 77 candidates each: 2.03 ms with 1, 1.89 ms with 2) */
#define DRAW_STAGE 64        /* keepers per particle collected in LDS */
#define DRAW_STAGE_STRIDE 68 /* words per particle there: 16-byte aligned rows, 8 distinct bank offsets */
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_draw_lists(const uint2* __restrict__ qpos, uint32_t n,
                                                          const uint32_t* __restrict__ nbr, uint32_t stride,
                                                          const uint32_t* __restrict__ cnt, uint32_t r2q,
                                                          uint32_t* __restrict__ nbr_out, uint32_t* __restrict__ cnt_out,
                                                          DevScalars* __restrict__ S) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[(SPH_BLOCK / 4) * DRAW_STAGE_STRIDE];
    /* the lists drawn here start their own travel bound */
    if (blockIdx.x == 0 && threadIdx.x == 0) S->disp_in = 0.0;
    const uint32_t q = threadIdx.x & 3u;
    const uint32_t i_raw = xcd_contiguous_block() * (SPH_BLOCK / 4) + (threadIdx.x >> 2);
    const bool active = i_raw < n;
    const uint32_t i = active ? i_raw : n - 1;
    const uint2 pi = qpos[i];
    const uint32_t c = active ? cnt[i] : 0u;
    const s16x2_t xyi = __builtin_bit_cast(s16x2_t, pi.x), zi = __builtin_bit_cast(s16x2_t, pi.y);
    const uint32_t* const row = nbr + (size_t)i * stride;
    uint32_t* const out = nbr_out + (size_t)i * stride;
    uint32_t* const mine = stage + (threadIdx.x >> 2) * DRAW_STAGE_STRIDE;
    const uint32_t last_group = stride - 4u;
    /* the bits of a sixteen-entry mask in front of this lane's four entries */
    const uint32_t front = (1u << (4u * q)) - 1u;
    uint32_t k = 0;
    for (uint32_t e0 = 0; e0 < c; e0 += 16u * DRAW_GROUPS) {
        uint32_t j[DRAW_GROUPS][4];
#pragma unroll
        for (int g = 0; g < DRAW_GROUPS; ++g) {
            const uint32_t eg = e0 + 16u * g + 4u * q;
            const u32x4_t v = *(const u32x4_t*)(row + (eg < last_group ? eg : last_group));
            j[g][0] = v.x; j[g][1] = v.y; j[g][2] = v.z; j[g][3] = v.w;
        }
        /* entries past the count are stale or zero but always index the arrays (the lists are
         * cleared when allocated): they are fetched like the others and dropped by the flags */
        uint2 pj[DRAW_GROUPS][4];
#pragma unroll
        for (int g = 0; g < DRAW_GROUPS; ++g)
#pragma unroll
            for (int m = 0; m < 4; ++m) /* a 32-bit byte offset from the array's base (the host checks the size) */
                pj[g][m] = *(const uint2*)((const char*)qpos + (j[g][m] << 3));
#pragma unroll
        for (int g = 0; g < DRAW_GROUPS; ++g) {
            /* how many of this lane's four entries lie below the count */
            const int room = (int)c - (int)(e0 + 16u * g + 4u * q);
            uint32_t nib = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                /* wrapped 16-bit differences; their squares add up to less than 2^32 */
                const s16x2_t dxy = __builtin_bit_cast(s16x2_t, pj[g][m].x) - xyi;
                int d2 = __builtin_amdgcn_sdot2(dxy, dxy, 0, false);
                if (DIM == 3) {
                    const s16x2_t dz = __builtin_bit_cast(s16x2_t, pj[g][m].y) - zi;
                    d2 = __builtin_amdgcn_sdot2(dz, dz, d2, false);
                }
                const bool keep = (m < room) & ((uint32_t)d2 <= r2q);
                nib |= keep ? 1u << m : 0u;
            }
            /* the quad's sixteen flags, in the row's order, in every lane */
            uint32_t w = nib << (4u * q);
            w |= (uint32_t)quad_perm_i<0xB1>((int)w); /* [1,0,3,2] */
            w |= (uint32_t)quad_perm_i<0x4E>((int)w); /* [2,3,0,1] */
            const uint32_t base = k + __popc(w & front);
            k += __popc(w);
            /* keepers to their slots, the others to a spare word behind the row's slots */
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t at = base + __popc(nib & ((1u << m) - 1u));
                mine[((nib >> m) & 1u) && at < DRAW_STAGE ? at : DRAW_STAGE + q] = j[g][m];
            }
            if (k > DRAW_STAGE) { /* a long row: what does not fit goes straight to memory */
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t at = base + __popc(nib & ((1u << m) - 1u));
                    if (((nib >> m) & 1u) && at >= DRAW_STAGE) out[at] = j[g][m];
                }
            }
        }
    }
    /* the collected keepers: whole groups of four, the tail padded with the particle itself */
    const uint32_t kk = k < DRAW_STAGE ? k : DRAW_STAGE;
    for (uint32_t e = 4u * q; e < kk; e += 16u) {
        u32x4_t t = *(const u32x4_t*)(mine + e);
        if (e + 1u >= kk) t.y = i;
        if (e + 2u >= kk) t.z = i;
        if (e + 3u >= kk) t.w = i;
        *(u32x4_t*)(out + e) = t;
    }
    if (active && q == 0) cnt_out[i] = k;
}

/* Four lanes per particle.  For every neighbouring column of cells (9 in 3-D, 3 in 2-D) the
 * quad reads ONE range of the cell table: the fine cells along the run axis that a sphere of
 * the list radius around the particle can reach in that column (none if the column itself is
 * out of reach), own particles and, in a slab run, the neighbours' particles.  It tests the
 * range sixteen candidates at a time on their 8-byte lattice entries (k_fill_qpos): a lane reads
 * its four consecutive entries with two 16-byte loads (the range is entered at an even index, so
 * the loads are aligned) and tests them with packed 16-bit subtractions and integer dot products;
 * the flags of the sixteen are combined across the quad into one mask per list in the range's
 * order, from which every lane counts the keepers in front of its own.  Keepers wait in a ring of
 * 32 words per list and particle in LDS and leave for memory sixteen at a time, as four aligned
 * 16-byte stores of the quad; the last group of a row is padded with the particle's own index
 * (the pair kernel fetches what a row names before it looks at the count).  A row lists its
 * candidates column by column (x slowest), own particles before halo particles, ascending along
 * the run axis.  (The version before this one -- 16-byte single-precision entries, two candidates
 * per lane and trip, a ballot and a scattered 4-byte store per keeper and list -- took 5.3 ms at
 * 12.7 M particles and 460 candidates tested per particle.) */
#define BUILD_RING 32
#define BUILD_STAGE_STRIDE (2 * BUILD_RING + 4) /* two rings and a spare word per lane */
typedef uint32_t u32x4_t_a16 __attribute__((ext_vector_type(4), aligned(16)));
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_build_list_ring(const Rec* __restrict__ rec, const uint2* __restrict__ qpos, uint32_t n,
                                                          GridParams g,
                                                          const uint32_t* __restrict__ cstart, double rl2, uint32_t rl2q,
                                                          uint32_t* __restrict__ nbr, uint32_t stride, uint32_t cap,
                                                          uint32_t* __restrict__ cnt, uint32_t* __restrict__ need,
                                                          int prune_walls, const uint32_t* __restrict__ hstart,
                                                          uint32_t rin2q, uint32_t* __restrict__ nbr_in,
                                                          uint32_t* __restrict__ cnt_in, const uint32_t* __restrict__ todo,
                                                          uint32_t todo_n, int stale_cells) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[(SPH_BLOCK / 4) * BUILD_STAGE_STRIDE];
    const uint32_t q = threadIdx.x & 3u;
    /* todo != nullptr: the rows of the todo_n particles named there (those k_build_list handed on);
     * otherwise the rows of all n particles */
    const uint32_t slot = (todo ? blockIdx.x : xcd_contiguous_block()) * (SPH_BLOCK / 4) + (threadIdx.x >> 2);
    const bool listed = todo ? slot < todo_n : slot < n;
    const uint32_t i_named = todo ? todo[listed ? slot : 0u] : slot;
    const bool active = listed && i_named < n;
    const uint32_t i = active ? i_named : n - 1;
    const uint2 pi = qpos[i];
    /* prune_walls: pairs of two wall particles are left out (see k_walls_at_rest): or-ed into the
     * squared distance, a wall candidate's mark puts it beyond any radius */
    const uint32_t wall_mask = (prune_walls && (pi.y & QPOS_WALL)) ? QPOS_WALL : 0u;
    const s16x2_t xyi = __builtin_bit_cast(s16x2_t, pi.x), zi = __builtin_bit_cast(s16x2_t, pi.y);
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi_d = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    const double run_i = DIM == 3 ? zi_d : yi;
    const double edge_d = 1.0 / g.inv[0]; /* all axes share one edge */
    const int cx = cell_coord(xi, g.lo[0], g.inv[0], g.n[0]);
    const int cy = DIM == 3 ? cell_coord(yi, g.lo[1], g.inv[1], g.n[1]) : 0;
    uint32_t* const row = nbr + (size_t)i * stride;
    /* nbr_in != nullptr: the step lists (the candidates within the smaller radius, same order)
     * are written as well, which saves the draw that would follow the rebuild */
    const bool two = nbr_in != nullptr;
    uint32_t* const row_in = two ? nbr_in + (size_t)i * stride : nullptr;
    uint32_t* const mine = stage + (threadIdx.x >> 2) * BUILD_STAGE_STRIDE;
    /* The particle's place inside its cell, in cell edges: the gap to a column on the low side
     * is fx, on the high side 1 - fx (never negative).  What a column can hold within reach lies
     * within sqrt(rl^2 - gap^2) of the particle along the run axis; the bound carries a margin
     * far above the rounding of fx and of the square root, so no candidate is lost. */
    const double fx = fmin(fmax((xi - g.lo[0]) * g.inv[0] - (double)cx, 0.0), 1.0);
    const double fy = DIM == 3 ? fmin(fmax((yi - g.lo[1]) * g.inv[1] - (double)cy, 0.0), 1.0) : 0.0;
    const double reach2 = rl2 * g.inv[0] * g.inv[0] * (1.0 + 1e-6);
    /* the bits of a sixteen-candidate mask in front of this lane's four */
    const uint32_t front = (1u << (4u * q)) - 1u;
    uint32_t k = 0, k_in = 0; /* keepers so far */
    uint32_t f = 0, f_in = 0; /* of them written to memory: multiples of 16 */
    /* sixteen keepers [f, f+16) of a ring leave for their row */
    auto flush16 = [&](const uint32_t* ring, uint32_t* dst_row, uint32_t done) {
        const u32x4_t_a16 t = *(const u32x4_t_a16*)(ring + (done & (BUILD_RING - 1u)) + 4u * q);
        if (done + 4u * q + 4u <= stride) *(u32x4_t_a16*)(dst_row + done + 4u * q) = t;
    };
    /* candidates [b, e) of one column; self_col: the range may hold the particle itself */
    auto walk = [&](uint32_t b, uint32_t e, bool self_col) {
        for (uint32_t j0 = b & ~1u; j0 < e; j0 += 16u) {
            const uint32_t jl = j0 + 4u * q; /* this lane's first candidate */
            /* in bounds of the array's allocation even past the last particle (see qpos_of) */
            const u32x4_t_a16 A = *(const u32x4_t_a16*)(qpos + jl), B = *(const u32x4_t_a16*)(qpos + jl + 2u);
            const uint32_t lo[4] = {A.x, A.z, B.x, B.z}, hi[4] = {A.y, A.w, B.y, B.w};
            uint32_t nib = 0, nib_in = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                /* wrapped 16-bit differences; their squares add up to less than 2^31 */
                const s16x2_t dxy = __builtin_bit_cast(s16x2_t, lo[m]) - xyi;
                uint32_t d2 = (uint32_t)__builtin_amdgcn_sdot2(dxy, dxy, 0, false);
                if (DIM == 3) {
                    const uint32_t dzw = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, hi[m]) - zi) & 0xFFFFu;
                    const s16x2_t dz = __builtin_bit_cast(s16x2_t, dzw);
                    d2 = (uint32_t)__builtin_amdgcn_sdot2(dz, dz, (int)d2, false);
                }
                d2 |= hi[m] & wall_mask;
                nib |= d2 <= rl2q ? 1u << m : 0u;
                nib_in |= d2 <= rin2q ? 1u << m : 0u;
            }
            /* candidates in front of the range (an odd b), behind it, and the particle itself */
            const int room = (int)e - (int)jl;
            uint32_t ok = room >= 4 ? 0xFu : (room > 0 ? (1u << room) - 1u : 0u);
            if (jl < b) ok &= ~1u;
            if (self_col || stale_cells) { /* stale_cells: the particle may have left the column it is filed under */
                const uint32_t s = i - jl;
                if (s < 4u) ok &= ~(1u << s);
            }
            if (!active) ok = 0;
            nib &= ok;
            nib_in &= nib;
            /* the quad's flags in the range's order, both lists in one word, in every lane */
            uint32_t w = (nib | (nib_in << 16)) << (4u * q);
            w |= (uint32_t)quad_perm_i<0xB1>((int)w); /* [1,0,3,2] */
            w |= (uint32_t)quad_perm_i<0x4E>((int)w); /* [2,3,0,1] */
            const uint32_t wl = w & 0xFFFFu, wi = w >> 16;
            const uint32_t base = k + __popc(wl & front), base_in = k_in + __popc(wi & front);
            k += __popc(wl);
            k_in += __popc(wi);
            /* keepers to their places in the rings, the others to this lane's spare word (what a
             * full row cannot take passes through the ring too and is dropped where it would be
             * stored: rows hold `stride` = `cap` entries) */
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t at = base + __popc(nib & ((1u << m) - 1u));
                mine[(nib >> m) & 1u ? (at & (BUILD_RING - 1u)) : 2 * BUILD_RING + q] = jl + (uint32_t)m;
            }
            if (k - f >= 16u) { /* at most 31 wait: the ring never wraps onto them */
                flush16(mine, row, f);
                f += 16u;
            }
            if (two) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t at = base_in + __popc(nib_in & ((1u << m) - 1u));
                    mine[(nib_in >> m) & 1u ? BUILD_RING + (at & (BUILD_RING - 1u)) : 2 * BUILD_RING + q] = jl + (uint32_t)m;
                }
                if (k_in - f_in >= 16u) {
                    flush16(mine + BUILD_RING, row_in, f_in);
                    f_in += 16u;
                }
            }
        }
    };
    /* the table entries of all neighbouring columns are requested together, then the ranges are walked */
    constexpr int NCOL = DIM == 3 ? 9 : 3;
    uint32_t cb[NCOL], ce[NCOL], qb[NCOL], qe[NCOL];
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        const int dx = DIM == 3 ? t / 3 - 1 : t - 1, dy = DIM == 3 ? t % 3 - 1 : 0;
        const int ax = cx + dx, ay = cy + dy;
        const double gx = dx < 0 ? fx : (dx > 0 ? 1.0 - fx : 0.0);
        const double gy = dy < 0 ? fy : (dy > 0 ? 1.0 - fy : 0.0);
        const double gap2 = gx * gx + gy * gy; /* in cell edges */
        const bool take = ax >= 0 && ax < g.n[0] && (DIM != 3 || (ay >= 0 && ay < g.n[1])) && !(gap2 > reach2);
        /* half-width of the reachable stretch of the column along the run axis */
        const double hw = sqrt(fmax(reach2 - gap2, 0.0)) * edge_d * (1.0 + 1e-6) + 1e-9 * edge_d;
        const uint32_t fa = (uint32_t)cell_coord(run_i - hw, g.lo[DIM - 1], g.invf, (int)g.nrun);
        const uint32_t fb = (uint32_t)cell_coord(run_i + hw, g.lo[DIM - 1], g.invf, (int)g.nrun);
        const uint32_t base = take ? column_index<DIM>(g, ax, ay) * g.nrun : 0u; /* column 0 always exists */
        const uint32_t b0 = cstart[base + fa], e0 = cstart[base + fb + 1u];
        cb[t] = take ? b0 : 0xFFFFFFFEu;
        ce[t] = take ? e0 : 0u;
        if (hstart) {
            const uint32_t b1 = hstart[base + fa], e1 = hstart[base + fb + 1u];
            qb[t] = take ? b1 : 0xFFFFFFFEu;
            qe[t] = take ? e1 : 0u;
        }
    }
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        /* own particles of the range, then (slab runs) the neighbours' particles in it */
        walk(cb[t], ce[t], t == NCOL / 2);
        if (hstart) walk(qb[t], qe[t], false);
    }
    /* what still waits in the rings: a last group each, padded with the particle itself */
    auto finish = [&](const uint32_t* ring, uint32_t* dst_row, uint32_t done, uint32_t total) {
        const uint32_t e = done + 4u * q;
        if (e < total && e + 4u <= stride) {
            u32x4_t_a16 t = *(const u32x4_t_a16*)(ring + (done & (BUILD_RING - 1u)) + 4u * q);
            if (e + 1u >= total) t.y = i;
            if (e + 2u >= total) t.z = i;
            if (e + 3u >= total) t.w = i;
            *(u32x4_t_a16*)(dst_row + e) = t;
        }
    };
    const uint32_t kc = k < cap ? k : cap, kc_in = k_in < cap ? k_in : cap;
    if (f < kc) finish(mine, row, f, kc);
    if (two && f_in < kc_in) finish(mine + BUILD_RING, row_in, f_in, kc_in);
    if (active && q == 0) {
        cnt[i] = kc;
        if (k > cap) atomicMax(need, k);
        if (two) cnt_in[i] = kc_in;
    }
}

/* The list build proper.  Same ranges, same tests and the same rows as k_build_list_ring above, but
 * the keepers are not placed one by one while the ranges are walked (two thirds of that kernel's
 * vector instructions: a rank, a ring slot and an LDS store per TESTED candidate and list, six out of
 * seven of them thrown away).  Walking a range only RECORDS, per sixteen candidates with a keeper
 * among them, where they start, the quad's two keep masks and how many keepers came before (12 bytes
 * in LDS, written by one lane).  When all ranges are done the rows are written from the records: a
 * lane takes a quarter of a row's groups of four, finds the record that holds its first keeper (a
 * binary search over the running counts), and from there peels the set bits off the masks -- work per
 * KEPT candidate -- into 16-byte stores.  Rows, counts and padding are exactly the ring kernel's.
 *   A quad has room for `trips_max` records (at most BUILD_TRIPS).  A particle that needs more (far
 * denser surroundings than the cells were sized for) is named in `todo` instead, counted in need[1],
 * and the host has k_build_list_ring write its rows.
 *   stale_cells: the cell table is that of an earlier sort and the particles have moved since (the
 * caller's range allows for it); a particle is then looked for in every column, not only in the one
 * it is in now, when it comes to leaving itself out of its row. */
#define BUILD_TRIPS 40
#define BUILD_REC_STRIDE (BUILD_TRIPS + 1)
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_build_list(const Rec* __restrict__ rec, const uint2* __restrict__ qpos, uint32_t n,
                                                          GridParams g,
                                                          const uint32_t* __restrict__ cstart, double rl2, uint32_t rl2q,
                                                          uint32_t* __restrict__ nbr, uint32_t stride, uint32_t cap,
                                                          uint32_t* __restrict__ cnt, uint32_t* __restrict__ need,
                                                          int prune_walls, const uint32_t* __restrict__ hstart,
                                                          uint32_t rin2q, uint32_t* __restrict__ nbr_in,
                                                          uint32_t* __restrict__ cnt_in, uint32_t trips_max,
                                                          uint32_t* __restrict__ todo, int stale_cells) {
    __shared__ uint2 s_rec[(SPH_BLOCK / 4) * BUILD_REC_STRIDE];   /* first of the sixteen; masks: list | step list << 16 */
    __shared__ uint32_t s_kb[(SPH_BLOCK / 4) * BUILD_REC_STRIDE]; /* keepers before the record, likewise packed */
    const uint32_t q = threadIdx.x & 3u;
    const uint32_t i_raw = xcd_contiguous_block() * (SPH_BLOCK / 4) + (threadIdx.x >> 2);
    const bool active = i_raw < n;
    const uint32_t i = active ? i_raw : n - 1;
    const uint2 pi = qpos[i];
    const uint32_t wall_mask = (prune_walls && (pi.y & QPOS_WALL)) ? QPOS_WALL : 0u;
    const s16x2_t xyi = __builtin_bit_cast(s16x2_t, pi.x), zi = __builtin_bit_cast(s16x2_t, pi.y);
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi_d = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    const double run_i = DIM == 3 ? zi_d : yi;
    const double edge_d = 1.0 / g.inv[0];
    const int cx = cell_coord(xi, g.lo[0], g.inv[0], g.n[0]);
    const int cy = DIM == 3 ? cell_coord(yi, g.lo[1], g.inv[1], g.n[1]) : 0;
    const bool two = nbr_in != nullptr;
    uint2* const myrec = s_rec + (threadIdx.x >> 2) * BUILD_REC_STRIDE;
    uint32_t* const mykb = s_kb + (threadIdx.x >> 2) * BUILD_REC_STRIDE;
    const double fx = fmin(fmax((xi - g.lo[0]) * g.inv[0] - (double)cx, 0.0), 1.0);
    const double fy = DIM == 3 ? fmin(fmax((yi - g.lo[1]) * g.inv[1] - (double)cy, 0.0), 1.0) : 0.0;
    const double reach2 = rl2 * g.inv[0] * g.inv[0] * (1.0 + 1e-6);
    const uint32_t lim = trips_max < BUILD_TRIPS ? trips_max : BUILD_TRIPS;
    uint32_t k = 0, k_in = 0; /* keepers so far */
    uint32_t T = 0;           /* records so far, counted on when they no longer fit */
    auto walk = [&](uint32_t b, uint32_t e, bool self_col) {
        for (uint32_t j0 = b & ~1u; j0 < e; j0 += 16u) {
            const uint32_t jl = j0 + 4u * q;
            const u32x4_t_a16 A = *(const u32x4_t_a16*)(qpos + jl), B = *(const u32x4_t_a16*)(qpos + jl + 2u);
            const uint32_t lo[4] = {A.x, A.z, B.x, B.z}, hi[4] = {A.y, A.w, B.y, B.w};
            uint32_t nib = 0, nib_in = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const s16x2_t dxy = __builtin_bit_cast(s16x2_t, lo[m]) - xyi;
                uint32_t d2 = (uint32_t)__builtin_amdgcn_sdot2(dxy, dxy, 0, false);
                if (DIM == 3) {
                    const uint32_t dzw = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, hi[m]) - zi) & 0xFFFFu;
                    const s16x2_t dz = __builtin_bit_cast(s16x2_t, dzw);
                    d2 = (uint32_t)__builtin_amdgcn_sdot2(dz, dz, (int)d2, false);
                }
                d2 |= hi[m] & wall_mask;
                nib |= d2 <= rl2q ? 1u << m : 0u;
                nib_in |= d2 <= rin2q ? 1u << m : 0u;
            }
            const int room = (int)e - (int)jl;
            uint32_t ok = room >= 4 ? 0xFu : (room > 0 ? (1u << room) - 1u : 0u);
            if (jl < b) ok &= ~1u;
            if (self_col || stale_cells) { /* stale_cells: the particle may have left the column it is filed under */
                const uint32_t s = i - jl;
                if (s < 4u) ok &= ~(1u << s);
            }
            if (!active) ok = 0;
            nib &= ok;
            nib_in &= nib;
            uint32_t w = (nib | (nib_in << 16)) << (4u * q);
            w |= (uint32_t)quad_perm_i<0xB1>((int)w);
            w |= (uint32_t)quad_perm_i<0x4E>((int)w);
            if (w != 0u) { /* the same in the quad's four lanes */
                if (q == 0 && T < lim) {
                    myrec[T] = make_uint2(j0, w);
                    mykb[T] = (k < 0xFFFFu ? k : 0xFFFFu) | ((k_in < 0xFFFFu ? k_in : 0xFFFFu) << 16);
                }
                ++T;
                k += __popc(w & 0xFFFFu);
                k_in += __popc(w >> 16);
            }
        }
    };
    constexpr int NCOL = DIM == 3 ? 9 : 3;
    uint32_t cb[NCOL], ce[NCOL], qb[NCOL], qe[NCOL];
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        const int dx = DIM == 3 ? t / 3 - 1 : t - 1, dy = DIM == 3 ? t % 3 - 1 : 0;
        const int ax = cx + dx, ay = cy + dy;
        const double gx = dx < 0 ? fx : (dx > 0 ? 1.0 - fx : 0.0);
        const double gy = dy < 0 ? fy : (dy > 0 ? 1.0 - fy : 0.0);
        const double gap2 = gx * gx + gy * gy;
        const bool take = ax >= 0 && ax < g.n[0] && (DIM != 3 || (ay >= 0 && ay < g.n[1])) && !(gap2 > reach2);
        const double hw = sqrt(fmax(reach2 - gap2, 0.0)) * edge_d * (1.0 + 1e-6) + 1e-9 * edge_d;
        const uint32_t fa = (uint32_t)cell_coord(run_i - hw, g.lo[DIM - 1], g.invf, (int)g.nrun);
        const uint32_t fb = (uint32_t)cell_coord(run_i + hw, g.lo[DIM - 1], g.invf, (int)g.nrun);
        const uint32_t base = take ? column_index<DIM>(g, ax, ay) * g.nrun : 0u;
        const uint32_t b0 = cstart[base + fa], e0 = cstart[base + fb + 1u];
        cb[t] = take ? b0 : 0xFFFFFFFEu;
        ce[t] = take ? e0 : 0u;
        if (hstart) {
            const uint32_t b1 = hstart[base + fa], e1 = hstart[base + fb + 1u];
            qb[t] = take ? b1 : 0xFFFFFFFEu;
            qe[t] = take ? e1 : 0u;
        }
    }
#pragma unroll
    for (int t = 0; t < NCOL; ++t) {
        walk(cb[t], ce[t], t == NCOL / 2);
        if (hstart) walk(qb[t], qe[t], false);
    }
    if (T > lim) { /* no room for this particle's records: its rows are written by the other kernel */
        if (active && q == 0) todo[atomicAdd(need + 1, 1u)] = i;
    } else {
        /* the rows from the records; sh: which half of the packed words */
        auto expand = [&](uint32_t sh, uint32_t total, uint32_t* dst_row) {
            const uint32_t kc = total < cap ? total : cap;
            const uint32_t G = (kc + 3u) >> 2, Gq = (G + 3u) >> 2; /* groups of four: in the row, per lane */
            const uint32_t g0 = q * Gq;
            const uint32_t g1 = g0 + Gq < G ? g0 + Gq : G;
            if (g0 < g1) { /* then kc > 0 and there is at least one record */
                const uint32_t r0 = 4u * g0; /* the rank of this lane's first keeper */
                uint32_t lo = 0, hi = T;
#pragma unroll
                for (int s = 0; s < 6; ++s) { /* the last record with no more than r0 keepers before it (2^6 > BUILD_TRIPS) */
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool le = ((mykb[mid] >> sh) & 0xFFFFu) <= r0;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid;
                }
                uint32_t t = lo;
                uint2 r = myrec[t];
                uint32_t m = (r.y >> sh) & 0xFFFFu;
                for (uint32_t d = r0 - ((mykb[t] >> sh) & 0xFFFFu); d > 0u && m != 0u; --d) m &= m - 1u;
                for (uint32_t gq = g0; gq < g1; ++gq) {
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t s = 0; s < 4u; ++s) {
                        v[s] = i; /* the last group's padding */
                        if (4u * gq + s < kc) {
                            while (m == 0u && t + 1u < T) {
                                ++t;
                                r = myrec[t];
                                m = (r.y >> sh) & 0xFFFFu;
                            }
                            const uint32_t bit = m != 0u ? (uint32_t)__builtin_ctz(m) : 0u;
                            v[s] = r.x + bit;
                            m &= m - 1u;
                        }
                    }
                    if (4u * gq + 4u <= stride) {
                        u32x4_t_a16 o;
                        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
                        *(u32x4_t_a16*)(dst_row + 4u * gq) = o;
                    }
                }
            }
        };
        if (active) {
            expand(0u, k, nbr + (size_t)i * stride);
            if (two) expand(16u, k_in, nbr_in + (size_t)i * stride);
            if (q == 0) {
                cnt[i] = k < cap ? k : cap;
                if (k > cap) atomicMax(need, k);
                if (two) cnt_in[i] = k_in < cap ? k_in : cap;
            }
        }
    }
}

/* Development check (SPH_CHECK_LISTS): every word of every row must name a particle slot, and no
 * count may exceed the rows' capacity; bad[0] counts the offenders. */
__global__ __launch_bounds__(SPH_BLOCK) void k_check_lists(const uint32_t* __restrict__ nbr, uint32_t stride, uint32_t cap,
                                                           const uint32_t* __restrict__ cnt, uint32_t n, uint32_t slots,
                                                           uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t wrong = cnt[i] > cap ? 1u : 0u;
    for (uint32_t e = 0; e < stride; ++e) wrong += nbr[(size_t)i * stride + e] >= slots ? 1u : 0u;
    if (wrong) atomicAdd(bad, wrong);
}

/* Diagnostic (SPH_TRACE): how well the rows of one list fill the pair kernel's waves.  A wave
 * takes 16 consecutive particles and runs max(ceil(count/4)) rounds.  out[0] += rounds summed
 * over the waves, out[1] += sum of ceil(count/4) over the particles, out[2] += entries,
 * out[3] += the rounds the waves would run if the 64 particles of a workgroup were dealt to its
 * four waves in the order of their row lengths. */
__global__ __launch_bounds__(SPH_BLOCK) void k_rounds_stats(const uint32_t* __restrict__ cnt, uint32_t n,
                                                            unsigned long long* __restrict__ out) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    const uint32_t c = i < n ? cnt[i] : 0u;
    uint32_t r = (c + 3u) >> 2, m = r;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 15u) == 0) atomicAdd(out, (unsigned long long)m);
    atomicAdd(out + 1, (unsigned long long)r);
    atomicAdd(out + 2, (unsigned long long)c);
    /* rank of this particle's row length among the 64 of its wave (ties by lane) */
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t o = (uint32_t)__shfl((int)r, l);
        rank += (o < r || (o == r && (uint32_t)l < lane)) ? 1u : 0u;
    }
    if ((rank & 15u) == 15u) atomicAdd(out + 3, (unsigned long long)r);
}

/* ------------------------------------------------------------------------------------------ */
/* the pair kernel                                                                            */

struct NState { /* state n, structure of arrays, cell order */
    const double* x[3];
    const double* v[3];
    const double* V[3];
    const double* f[3];
    const float* rho;
    const float* Rrho;
    const float* p; /* only written: by the sweep that also commits the step */
};

struct OutState { /* f, V, Rrho produced by Forces */
    double* f[3];
    double* V[3];
    float* Rrho;
    /* a fused sweep whose new positions the step lists are drawn on next also leaves their lattice
     * entries (what k_fill_qpos would make of its records); nullptr otherwise */
    uint2* qpos;
    QLattice lat;
};

/* (rho/rho0)^gam - 1 as the reference's pow(rho/rho0, 7.0) - 1 evaluates it: the power is
 * rounded to double once and 1 is subtracted exactly.  The power is formed from
 * d = x - 1 (exact for x near 1) so that its rounding matches a correctly rounded pow. */
__device__ __forceinline__ double tait_pow_m1(double x, double gam) {
    if (gam == 7.0) {
        const double d = x - 1.0;
        const double e = d * (7.0 + d * (21.0 + d * (35.0 + d * (35.0 + d * (21.0 + d * (7.0 + d))))));
        const double pw = 1.0 + e;
        return pw - 1.0;
    }
    return pow(x, gam) - 1.0;
}

/* 1/x for a normal positive x, to about one unit in the last place: the hardware estimate
 * plus two Newton steps. */
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}
/* x^(-1/2) for a normal positive x */
__device__ __forceinline__ double fast_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    double e = fma(-hx * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-hx * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

/* sqrt(x) for x >= 0 (zero or normal): x * x^(-1/2) from the hardware estimate refined once,
 * then one correction of the root itself; within one unit in the last place.  The added
 * 1e-300 changes no x that matters and keeps x = 0 (a particle and itself, or two particles at
 * one point) away from 0 * inf. */
__device__ __forceinline__ double fast_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x + 1e-300);
    const double e = fma(-(0.5 * x) * y, y, 0.5);
    y = fma(y, e, y);
    const double r = x * y;
    return fma(fma(-r, r, x), 0.5 * y, r);
}

/* Streamed-once data (candidate rows, state n, everything written) is moved with the
 * non-temporal hint, so that the L2 keeps what is reused: the neighbour records. */
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
    return *p; /* measured: the non-temporal hint on loads buys no time and costs 30 % more
                  fetched bytes (a row's second half is fetched again) */
}
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) { __builtin_nontemporal_store(v, p); }

/* a[q] for q in 0..2 by selects (a lane-dependent index into an argument array would go
 * through memory) */
template <typename T>
__device__ __forceinline__ T* pick3(T* const (&a)[3], int q) {
    T* const a0 = a[0];
    T* const a1 = a[1];
    T* const a2 = a[2];
    return q == 1 ? a1 : (q == 2 ? a2 : a0);
}
/* the low (which = 0) or high float of an 8-byte pair that was loaded as a double */
__device__ __forceinline__ float half_of(double pair, uint32_t which) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(pair);
    return __uint_as_float(which ? (uint32_t)(b >> 32) : (uint32_t)b);
}

/* Hooks for timing experiments (tools/pair_hooks.h, given to the compiler with -include by the
 * tools that use them); they are empty in the library. */
#ifndef PAIR_HOOK_FETCH
#define PAIR_HOOK_FETCH(j, i)
#endif
#ifndef PAIR_HOOK_HIT
#define PAIR_HOOK_HIT(hit, t)
#endif

#ifndef PAIR_WAVES
#define PAIR_WAVES 4 /* resident waves per SIMD the register budget is set for */
#endif
#ifndef PAIR_BLOCK
#define PAIR_BLOCK 256 /* threads per workgroup of the pair kernel */
#endif
#define PAIR_PARTICLES (SPH_BLOCK / 4)  /* particles per workgroup of the list kernels (four lanes each) */
#define PAIR_GROUP 16                   /* particles a wave has in its rounds at a time: four lanes each */
#define PAIR_GROUPS 4                   /* groups a wave works through: 64 particles, one per lane at the end */
#define PAIR_WAVE_SPAN (PAIR_GROUP * PAIR_GROUPS)
#define PAIR_SPAN ((PAIR_BLOCK / 64) * PAIR_WAVE_SPAN) /* particles per workgroup of the pair kernel */
/* LDS transpose area: per wave four 1-KiB blocks (one per record of a quad's round); block r
 * is skewed by r*16 bytes so that the four lanes of a quad read four different 16-byte
 * bank groups */
#define PAIR_LDS_BLOCK (1024 + 64)
#define PAIR_LDS_WAVE (4 * PAIR_LDS_BLOCK)
/* per wave: what the rounds leave for the finish, PAIR_KEEP doubles for each of its 64 particles */
#define PAIR_KEEP 11

/* A wave takes 64 consecutive particles of state n+1 (cell order).
 *
 *  Forces (WCXSPH3D.cpp:139-205), four lanes per particle, 16 particles (a group) at a time, four
 *  groups one after the other.  Each lane of a quad walks every fourth entry of the particle's
 *  candidate list.  The four records a quad needs per round are fetched by the quad together --
 *  lane q loads bytes 16q..16q+15 of each -- so one wave instruction touches 16 records of 64
 *  contiguous bytes instead of 64 scattered 16-byte pieces (the vector memory pipe is paced by
 *  distinct segments per instruction).  The pieces are exchanged through a per-wave LDS area:
 *  copied there lane-linear straight from memory (global->LDS loads, no register stop), read
 *  back one whole record per lane.  The copies of round t+1 are in flight during the arithmetic
 *  of round t.  Pairs with dist2_exact < search_radius contribute momentum, continuity, XSPH and
 *  max(mu); the four partial sums of a quad are added when the group's rounds are over and left
 *  in LDS with the particle's position.
 *
 *  The finish, one lane per particle (lane 4p+q takes particle 16q+p of the wave's 64): gravity,
 *  the wall reset and, FUSED, the Newmark-beta update, the Tait pressure and the position-change
 *  sum (WCXSPH3D.cpp:260-285).  In this form a wave instruction serves 64 particles instead of
 *  16, every array of state n is read as one contiguous run with the array's base on the scalar
 *  side (no per-lane choice of pointers), and the new records go out as whole kilobytes: they
 *  are laid into the transpose area and stored lane-linear from there.
 */
template <int DIM, bool XSPH, bool FUSED, bool WIDE>
__global__ __launch_bounds__(PAIR_BLOCK, PAIR_WAVES) void k_pair(DevParams P, const Rec* __restrict__ rin,
                                                    const uint32_t* __restrict__ nbr, uint32_t stride,
                                                    const uint32_t* __restrict__ cnt, OutState out, int store_out,
                                                    int zero_wall, NState ns, const uint32_t* __restrict__ id,
                                                    Rec* __restrict__ rout, float* __restrict__ pout,
                                                    DevScalars* __restrict__ S, double* __restrict__ part,
                                                    uint32_t first, uint32_t count, uint32_t part_base,
                                                    uint32_t part_n) {
    __shared__ __attribute__((aligned(16))) char stage[(PAIR_BLOCK / 64) * PAIR_LDS_WAVE];
    __shared__ double keep[PAIR_BLOCK / 64][PAIR_KEEP][PAIR_WAVE_SPAN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 3, p = lane >> 2;
    /* Workgroups are dealt round-robin to the eight XCDs, each with its own L2; see
     * xcd_contiguous_block.  (Placement only changes speed, never results.) */
    const uint32_t block = xcd_contiguous_block(); /* the grid is a multiple of 8 */
    /* this launch covers particles [first, end) (a slab run launches its boundary layers and its
     * interior separately); the wave's 64 start at wbase */
    const uint32_t end = first + count;
    const uint32_t wbase = first + block * PAIR_SPAN + (uint32_t)wave * PAIR_WAVE_SPAN;
    char* const wstage = stage + wave * PAIR_LDS_WAVE;
    double (*const wkeep)[PAIR_WAVE_SPAN] = keep[wave];

    const double sqrtBgam = P.sqrtBgam;
    const double invH = P.invH;
    const double gconst = P.gconst;
    /* While the record array is smaller than 4 GiB (!WIDE) a 32-bit byte offset from its base
     * addresses any piece: one shift-or per load instead of 64-bit arithmetic. */
    const uint32_t qoff = 16u * (uint32_t)q;
    const uint32_t last_group = stride - 4u;
    uint32_t hits = 0; /* wave total, kept on the scalar side */
    /* the time step of this sweep: read once, ahead of everything, and kept on the scalar side */
    double dt = 0.0;
    /* store_out == 2: the step's last sweep in a single domain also makes state n+1 the new state n
     * (pn = pnp1, WCXSPH3D.cpp:301) unless the step has to be made again.  No earlier sweep of a step
     * is left unchecked and the last one is not checked at all, so the flag is final when this
     * launch starts.  A particle's state n is read by its own lane only, before that lane writes it.
     * f, V and Rrho are then written once, as state n's; the host copies them into state n+1's
     * arrays when somebody asks for those. */
    bool commit = false;
    if (FUSED) {
        const double v = S->dt;
        dt = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
        commit = store_out == 2 && __builtin_amdgcn_readfirstlane(S->violation) == 0;
    }

    for (uint32_t g = 0; g < PAIR_GROUPS; ++g) {
        const uint32_t gbase = wbase + g * PAIR_GROUP;
        if (gbase >= end) break; /* wave-uniform */
        const uint32_t i_raw = gbase + (uint32_t)p;
        const bool active = i_raw < end;
        const uint32_t i = active ? i_raw : end - 1;

        /* The particle's own record: lane q loads chunk q (the wave reads 16 whole records, 1 KiB in
         * a row) and the quad passes the pieces round. */
        const f64x2_t own = *((const f64x2_t*)(rin + i) + q);
        const double xi0 = quad_perm_d<0x00>(own.x), vi0 = quad_perm_d<0x00>(own.y);
        const double xi1 = quad_perm_d<0x55>(own.x), vi1 = quad_perm_d<0x55>(own.y);
        const double xi2 = quad_perm_d<0xAA>(own.x), vi2 = quad_perm_d<0xAA>(own.y);
        const double rho_m = quad_perm_d<0xFF>(own.x); /* the floats rho and m */
        const double prr_i = quad_perm_d<0xFF>(own.y);
        const float rho_if = half_of(rho_m, 0u), m_if = half_of(rho_m, 1u);
        const double rho_i = (double)rho_if;
        const double c_i = sqrtBgam * fast_rsq(rho_i); /* sqrt(B*gam/rho), WCXSPH3D.cpp:172 */

        /* The candidate row of particle i: entry e at nbr[i*cap + e].  Every fourth round each
         * lane fetches 16 bytes of it: lane q holds the four candidates of round 4u+q.
         * Unconditional and in bounds; entries past the count are replaced where they are used. */
        const uint32_t* const row = nbr + (size_t)i * stride;
        auto fetch_vec = [&](uint32_t u) -> uint4 {
            const uint32_t e = u * 16u + (uint32_t)q * 4u;
            const u32x4_t v = stream_load((const u32x4_t*)(row + (e < last_group ? e : last_group)));
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        /* the count and the head of the candidate row are requested together with the record */
        const uint32_t c = active ? stream_load(cnt + i) : 0u;
        const uint4 vhead = fetch_vec(0);
        const uint4 vhead1 = fetch_vec(1);
        /* rounds of four candidates; the trip count is made wave-uniform */
        const uint32_t rounds = __builtin_amdgcn_readfirstlane(quads_allmax((c + 3u) >> 2));

        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
        double Vs0 = 0.0, Vs1 = 0.0, Vs2 = 0.0;
        double rr = 0.0, mumax = 0.0;

        /* The pieces go from memory into the transpose area without a stop in registers: a 16-byte
         * global->LDS copy, every lane naming its own source, the destination the wave-uniform base
         * plus lane*16 -- the lane-linear image the transpose reads.  The copies of round t+1 are
         * issued once round t's records have been read out of the area (lgkmcnt) and are waited for
         * (vmcnt) a whole round of arithmetic later. */
        auto piece = [&](uint32_t j) -> const char* {
            PAIR_HOOK_FETCH(j, i)
            return !WIDE ? (const char*)rin + ((j << 6) | qoff) : (const char*)(rin + j) + 16 * q;
        };
        auto copy16 = [&](const char* src, char* dst_wave_base) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
        };
        auto load4 = [&](uint32_t j0, uint32_t j1, uint32_t j2, uint32_t j3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the area has been read */
            copy16(piece(j0), wstage);
            copy16(piece(j1), wstage + 1 * PAIR_LDS_BLOCK + 16);
            copy16(piece(j2), wstage + 2 * PAIR_LDS_BLOCK + 32);
            copy16(piece(j3), wstage + 3 * PAIR_LDS_BLOCK + 48);
        };
        /* load round t, whose four candidates sit in lane (t & 3) of the quad */
#define LOAD_FROM(VEC, R, T)                                                        \
        {                                                                               \
            const uint32_t a0 = (uint32_t)quad_perm_i<(R)*0x55>((int)(VEC).x);           \
            const uint32_t a1 = (uint32_t)quad_perm_i<(R)*0x55>((int)(VEC).y);           \
            const uint32_t a2 = (uint32_t)quad_perm_i<(R)*0x55>((int)(VEC).z);           \
            const uint32_t a3 = (uint32_t)quad_perm_i<(R)*0x55>((int)(VEC).w);           \
            /* a quad whose row ends before this round fetches nothing (a wave runs as many rounds  \
             * as its longest row needs); within a row's last round the entries past the count are  \
             * stale or zero but always index the arrays (the lists are cleared when allocated),   \
             * and what is fetched for them is never used */                                        \
            if (4u * (T) < c) load4(a0, a1, a2, a3);                                                \
        }
        auto exchange = [&]() -> Rec {
            /* the copies left a round ago; the vector-memory counter also covers them */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return *(const Rec*)(wstage + q * PAIR_LDS_BLOCK + q * 16 + p * 64);
        };

        const int my_limit = (int)c - q; /* lane q holds entry 4t+q of round t: valid while 4t < c-q */
        auto pair_term = [&](uint32_t t, const Rec& rj) {
            const bool valid = (int)(t * 4u) < my_limit;
            const double Rx = rj.xv[0][0] - xi0, Ry = rj.xv[1][0] - xi1;
            const double Rz = DIM == 3 ? rj.xv[2][0] - xi2 : 0.0;
            const double d2 = dist2_exact(Rx, Ry, Rz);
            bool hit = valid && d2 < P.R2;
            PAIR_HOOK_HIT(hit, t)
            hits += (uint32_t)__popcll(__ballot(hit));
            if (hit) {
                const double Vx = rj.xv[0][1] - vi0, Vy = rj.xv[1][1] - vi1;
                const double Vz = DIM == 3 ? rj.xv[2][1] - vi2 : 0.0;
                const double r = fast_sqrt(d2);
                const double qq = r * invH;
                /* W2Kernel / W2GradK vanish for q >= 2 (WCXSPH3D.cpp:124,133): the clamp makes the
                 * factor (1 - q/2) an exact zero there */
                const double t1 = fmax(1.0 - 0.5 * qq, 0.0);
                const double t2 = t1 * t1;
                const double gs = gconst * (t2 * t1);
                const double Kern = XSPH ? (t2 * t2) * (2.0 * qq + 1.0) * P.correc : 0.0;
                const double gx = Rx * gs, gy = Ry * gs, gz = Rz * gs;
                const double rho_j = (double)rj.rho;
                const double cbar = 0.5 * (c_i + sqrtBgam * fast_rsq(rho_j));
                double vdotr = Vx * Rx + Vy * Ry;
                if (DIM == 3) vdotr += Vz * Rz;
                const double rhoij = 0.5 * (double)(rho_if + rj.rho); /* float add, as the reference */
                const double den = d2 + P.hvisc;
                const double iw = fast_rcp(den * rhoij);
                const double inv_rho = iw * den;
                const double muij = P.H * vdotr * (iw * rhoij);
                mumax = fmax(mumax, muij);
                double vg = Vx * gx + Vy * gy;
                if (DIM == 3) vg += Vz * gz;
                if (XSPH) {
                    /* viscosity only for approaching pairs (WCXSPH3D.cpp:175-178): mu has the sign
                     * of vdotr, and a zero factor gives the same zero */
                    const double pifac = P.alpha * cbar * fmin(muij, 0.0) * inv_rho;
                    const double mj = (double)rj.m;
                    const double sc = mj * (pifac - prr_i - rj.prr);
                    acc0 += gx * sc;
                    acc1 += gy * sc;
                    if (DIM == 3) acc2 += gz * sc;
                    const double xs = P.eps * (mj * inv_rho) * Kern;
                    Vs0 += xs * Vx;
                    Vs1 += xs * Vy;
                    if (DIM == 3) Vs2 += xs * Vz;
                    rr -= mj * vg;
                } else {
                    const double pifac = -P.alpha * cbar * fmin(muij, 0.0) * inv_rho;
                    const double sc = pifac + prr_i + rj.prr;
                    acc0 -= gx * sc;
                    acc1 -= gy * sc;
                    rr -= vg;
                }
            }
        };

        /* Round t: pass its pieces through LDS, start the loads of round t+1, then do the
         * arithmetic of round t while they fly.  R = t & 3 is spelled out because the lane
         * pattern of a cross-lane move is part of the instruction. */
#define ROUND(R, NEXTVEC, NEXTR)                               \
        {                                                          \
            const uint32_t t = u * 4u + (R);                       \
            if (t >= rounds) break;                                \
            const Rec rj = exchange();                             \
            if (t + 1 < rounds) LOAD_FROM(NEXTVEC, NEXTR, t + 1)   \
            pair_term(t, rj);                                      \
        }

        if (rounds > 0) {
            /* the entries of rounds 0..7 were requested with the rest up front; those of super-round
             * u+2 are requested at the top of super-round u, so that the wait for a round's records
             * (the counter is in order) never includes a row fetch that has just left */
            uint4 vcur = vhead, vnext = vhead1;
            LOAD_FROM(vcur, 0, 0u)
            const uint32_t nsup = (rounds + 3u) >> 2;
            for (uint32_t u = 0; u < nsup; ++u) {
                uint4 vafter = vnext;
                ROUND(0, vcur, 1)
                /* behind round 1's copies, so a round old at the next wait */
                if (u + 2 < nsup) vafter = fetch_vec(u + 2);
                ROUND(1, vcur, 2)
                ROUND(2, vcur, 3)
                ROUND(3, vnext, 0)
                vcur = vnext;
                vnext = vafter;
            }
        }
#undef ROUND
#undef LOAD_FROM

        /* quad totals, left for the finish by the quad's first lane: particle s = 16g+p of the wave */
        acc0 = quad_sum(acc0); acc1 = quad_sum(acc1);
        if (DIM == 3) acc2 = quad_sum(acc2);
        if (XSPH) {
            Vs0 = quad_sum(Vs0); Vs1 = quad_sum(Vs1);
            if (DIM == 3) Vs2 = quad_sum(Vs2);
        }
        rr = quad_sum(rr);
        mumax = quad_max(mumax);
        const float Rrho_f = XSPH ? (float)rr : (float)(rr * P.mass);
        if (q == 0) {
            const uint32_t s = g * PAIR_GROUP + (uint32_t)p;
            wkeep[0][s] = acc0;
            wkeep[1][s] = acc1;
            wkeep[2][s] = acc2;
            wkeep[3][s] = vi0 + Vs0; /* XSPH: V = v + eps*sum (WCXSPH3D.cpp:149,188) */
            wkeep[4][s] = vi1 + Vs1;
            wkeep[5][s] = vi2 + Vs2;
            wkeep[6][s] = __longlong_as_double(
                (long long)(((unsigned long long)__float_as_uint(m_if) << 32) | (unsigned long long)__float_as_uint(Rrho_f)));
            wkeep[7][s] = mumax;
            wkeep[8][s] = xi0;
            wkeep[9][s] = xi1;
            wkeep[10][s] = xi2;
        }
    }

    /* ---- the finish: lane 4p+q takes particle s = 16q+p of the wave's 64 ---- */
    const uint32_t s = (uint32_t)q * PAIR_GROUP + (uint32_t)p;
    const uint32_t e_raw = wbase + s;
    const bool mine = e_raw < end;
    const uint32_t e = mine ? e_raw : end - 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* what the quads left is in place */
    double fd[3], Vd[3], xo[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        fd[d] = wkeep[d][s];
        Vd[d] = wkeep[3 + d][s];
        xo[d] = wkeep[8 + d][s];
    }
    const double rm = wkeep[6][s];
    const float Rrho_f = half_of(rm, 0u), m_f = half_of(rm, 1u);
    const double mumax = mine ? wkeep[7][s] : 0.0;
    const bool wall = zero_wall && id[e] < P.bound_parts;
    double f2 = 0.0, move2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        if (!XSPH) fd[d] *= P.mass;
        if (d == P.g_axis) fd[d] += P.g;
        if (wall) { /* WCXSPH3D.cpp:260-264 */
            fd[d] = 0.0;
            Vd[d] = 0.0;
        }
        if (mine && store_out == 1) {
            stream_store(out.f[d] + e, fd[d]);
            if (XSPH) stream_store(out.V[d] + e, Vd[d]);
        }
    }
    if (mine && store_out == 1) stream_store(out.Rrho + e, Rrho_f);
    if (FUSED) {
        /* WCXSPH3D.cpp:271-277 */
        const double hdt = 0.5 * dt;
        const double dt2 = dt * dt;
        const double c1 = 0.5 * dt2 * P.omb;
        const double c2 = dt2 * P.beta;
        f64x2_t w[4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            w[d].x = 0.0; /* the unused component of a 2-D run stays zero */
            w[d].y = 0.0;
            if (d < DIM) {
                const double fn = stream_load(ns.f[d] + e);
                const double vn = stream_load(ns.v[d] + e);
                const double adv = XSPH ? stream_load(ns.V[d] + e) : vn;
                const double xn = stream_load(ns.x[d] + e);
                const double xnew = xn + dt * adv + c1 * fn + c2 * fd[d];
                w[d].x = xnew;
                w[d].y = vn + hdt * (fn + fd[d]);
                const double ex = xnew - xo[d];
                f2 += fd[d] * fd[d];
                move2 += ex * ex;
                if (commit && mine) {
                    stream_store(const_cast<double*>(ns.x[d]) + e, xnew);
                    stream_store(const_cast<double*>(ns.v[d]) + e, w[d].y);
                    stream_store(const_cast<double*>(ns.f[d]) + e, fd[d]);
                    if (XSPH) stream_store(const_cast<double*>(ns.V[d]) + e, Vd[d]);
                }
            }
        }
        const float Rrho_n = stream_load(ns.Rrho + e), rho_n = stream_load(ns.rho + e);
        const float rsum = Rrho_n + Rrho_f; /* float add, as the reference */
        const float rho_new = (float)((double)rho_n + hdt * (double)rsum);
        const float p_new = (float)(P.B * tait_pow_m1((double)rho_new / P.rho0, P.gam));
        w[3].x = __longlong_as_double(
            (long long)(((unsigned long long)__float_as_uint(m_f) << 32) | (unsigned long long)__float_as_uint(rho_new)));
        w[3].y = (double)p_new / ((double)rho_new * (double)rho_new);
        if (mine) stream_store(pout + e, p_new);
        if (out.qpos != nullptr && mine) out.qpos[e] = lattice_entry(out.lat, DIM, w[0].x, w[1].x, w[2].x);
        if (commit && mine) {
            stream_store(const_cast<float*>(ns.rho) + e, rho_new);
            stream_store(const_cast<float*>(ns.p) + e, p_new);
            stream_store(const_cast<float*>(ns.Rrho) + e, Rrho_f);
        }
        /* The new records leave as whole kilobytes: particle s lays its four chunks into the
         * transpose area (block s/16, record s%16), and the wave stores each block lane-linear --
         * 16 whole records per instruction. */
        char* const slot = wstage + q * PAIR_LDS_BLOCK + p * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f64x2_t*)(slot + 16 * k) = w[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < PAIR_GROUPS; ++b) {
            const uint32_t t_raw = wbase + (uint32_t)b * PAIR_GROUP + (uint32_t)p; /* record p of block b, chunk q */
            const f64x2_t v = *(const f64x2_t*)(wstage + b * PAIR_LDS_BLOCK + lane * 16);
            if (t_raw < end) stream_store((f64x2_t*)(rout + t_raw) + q, v);
        }
        if (!mine) {
            f2 = 0.0;
            move2 = 0.0;
        }
    }

    /* block reductions: max(mu), max |f|^2, max move^2, sum move^2, pair count */
    __shared__ double red[PAIR_BLOCK / 64][5];
    const double wm = wave_allmax(mumax);
    if (FUSED) {
        const double wf = wave_allmax(f2);
        const double wv = wave_allmax(move2 == move2 ? move2 : 1e300); /* NaN counts as huge */
        const double ws = wave_allsum(move2);
        const double wh = (double)hits;
        if (lane == 0) {
            red[wave][0] = wm; red[wave][1] = wf; red[wave][2] = wv; red[wave][3] = ws; red[wave][4] = wh;
        }
    } else if (lane == 0) {
        red[wave][0] = wm;
    }
    __syncthreads();
    /* no atomics: with tens of thousands of workgroups, adds to one word serialise at the
     * memory side.  Each workgroup leaves its five partials; two small kernels fold them. */
    if (threadIdx.x == 0) {
        double a = red[0][0];
        for (int w = 1; w < PAIR_BLOCK / 64; ++w) a = fmax(a, red[w][0]);
        const uint32_t nb = part_n;
        part += part_base;
        part[block] = a;
        if (FUSED) {
            double b = red[0][1], c2 = red[0][2], sm = red[0][3], h = red[0][4];
            for (int w = 1; w < PAIR_BLOCK / 64; ++w) {
                b = fmax(b, red[w][1]);
                c2 = fmax(c2, red[w][2]);
                sm += red[w][3];
                h += red[w][4];
            }
            part[nb + block] = b;
            part[2 * (size_t)nb + block] = c2;
            part[3 * (size_t)nb + block] = sm;
            part[4 * (size_t)nb + block] = h;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* scalars                                                                                    */

#define PART_FIELDS 5   /* max mu, max |f|^2, max move^2, sum move^2, pair count */
#define PART_GROUPS 64

/* First fold: group g (blockIdx.x) reduces a fixed contiguous share of the workgroup partials of
 * field blockIdx.y, in a fixed order (strided per thread, butterfly per wave, waves in turn), so
 * the sums never depend on scheduling.  part2[field*PART_GROUPS + g]. */
__global__ __launch_bounds__(SPH_BLOCK) void k_fold_parts(const double* __restrict__ part, uint32_t nb, int nfields,
                                                          double* __restrict__ part2) {
    __shared__ double sh[SPH_BLOCK / 64];
    const int f = blockIdx.y;
    const bool is_sum = f >= 3;
    const uint32_t share = (nb + PART_GROUPS - 1) / PART_GROUPS;
    const uint32_t lo = blockIdx.x * share, hi = min(nb, lo + share);
    const double* const src = part + (size_t)f * nb;
    double v = 0.0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += SPH_BLOCK) {
        const double x = src[i];
        v = is_sum ? v + x : fmax(v, x);
    }
    v = is_sum ? wave_sum(v) : wave_max(v);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sh[0];
        for (int w = 1; w < SPH_BLOCK / 64; ++w) a = is_sum ? a + sh[w] : fmax(a, sh[w]);
        part2[f * PART_GROUPS + blockIdx.x] = a;
    }
}

__device__ __forceinline__ double fold_groups(const double* part2, int f) {
    double v = 0.0;
    for (int g = 0; g < PART_GROUPS; ++g) v = f >= 3 ? v + part2[f * PART_GROUPS + g] : fmax(v, part2[f * PART_GROUPS + g]);
    return v;
}

/* After Forces alone: maxmu (WCXSPH3D.cpp:141,198-200). */
__global__ void k_forces_finalize(DevScalars* S, const double* __restrict__ part2) { S->maxmu = fold_groups(part2, 0); }

/* Start of a Newmark_Beta call. */
__global__ void k_step_begin(DevScalars* S) { S->violation = 0; }

/* End of sweep k: errsum, logbase, the travel bounds that guard the lists, and after the last
 * sweep the new time step (WCXSPH3D.cpp:280-288, 293-298).
 *
 * The positions just written feed the next sweep's pair loop through lists built earlier.  A
 * list drawn with radius r' > R is complete for the search radius R while no particle has
 * travelled more than (r' - R)/2 since it was drawn.  Two bounds are kept: disp since the
 * candidate lists were built, disp_in since the step lists were drawn (by the step's second
 * sweep, from the candidates).  cand_limit >= 0: the next sweep walks the candidate lists and
 * needs disp <= cand_limit.  check_in: the next sweep walks the step lists and needs
 * disp_in <= limit_in.  reset_in: the next sweep draws the step lists, their bound starts at 0. */
__global__ void k_sweep_finalize(DevParams P, DevScalars* S, const double* __restrict__ part2, int k, int last,
                                 double cand_limit, int check_in, int reset_in) {
    const double maxmu = fold_groups(part2, 0);
    const double maxf2 = fold_groups(part2, 1);
    const double move2 = fold_groups(part2, 2);
    const double errsum = fold_groups(part2, 3);
    S->hit_pairs = (unsigned long long)fold_groups(part2, 4);
    S->errsum = errsum;
    if (k == 0) {
        S->errsum0 = errsum;
        S->logbase = log10(sqrt(errsum / P.npts_d));
    }
    S->maxmu = maxmu;
    const double move = sqrt(move2);
    S->disp += move;
    const double din = reset_in ? 0.0 : S->disp_in + move;
    S->disp_in = din;
    if (!last && cand_limit >= 0.0 && !(S->disp <= cand_limit)) S->violation = 1;
    if (!last && check_in && !(din <= S->limit_in)) S->violation = 1;
    if (last) {
        const double maxf = sqrt(maxf2);
        S->maxf = maxf;
        const double dtf = sqrt(P.H / maxf);
        const double dtcv = P.H / (P.Cs + maxmu);
        S->dt_next = 0.3 * fmin(dtf, dtcv);
    }
}

/* pn = pnp1 and t += dt (WCXSPH3D.cpp:301-302), skipped when the step has to be redone.  Slab
 * runs (a single domain's last sweep writes state n itself): x, v and rho from the records; f, V,
 * Rrho and p become state n's by the host exchanging the arrays once it has seen that the step
 * stands. */
struct CommitArgs {
    uint32_t n;
    int dim;
    const Rec* rec;
    double* nx[3];
    double* nv[3];
    float* nrho;
};

__global__ __launch_bounds__(SPH_BLOCK) void k_commit(CommitArgs a, DevScalars* S) {
    if (S->violation) return;
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const Rec r = a.rec[i];
    for (int d = 0; d < a.dim; ++d) {
        a.nx[d][i] = r.xv[d][0];
        a.nv[d][i] = r.xv[d][1];
    }
    a.nrho[i] = r.rho;
}

__global__ void k_commit_scalars(DevScalars* S) {
    if (S->violation) return;
    S->dt = S->dt_next;
    S->t = (float)((double)S->t + S->dt); /* float t += double dt */
}

/* ------------------------------------------------------------------------------------------ */
/* state plumbing                                                                             */

/* state n (SoA) -> pair-loop records */
struct PackArgs {
    uint32_t n;
    int dim;
    const double* x[3];
    const double* v[3];
    const float* rho;
    const float* p;
    const float* m;
    Rec* rec;
};
__global__ __launch_bounds__(SPH_BLOCK) void k_pack(PackArgs a) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    Rec r;
    for (int d = 0; d < 3; ++d) {
        r.xv[d][0] = d < a.dim ? a.x[d][i] : 0.0;
        r.xv[d][1] = d < a.dim ? a.v[d][i] : 0.0;
    }
    r.rho = a.rho[i];
    r.m = a.m[i];
    r.prr = (double)a.p[i] / ((double)r.rho * (double)r.rho);
    a.rec[i] = r;
}

/* after rho or p of state n+1 changed from outside: re-form p/rho^2 */
__global__ __launch_bounds__(SPH_BLOCK) void k_fix_prr(Rec* __restrict__ rec, const float* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double rho = (double)rec[i].rho;
    rec[i].prr = (double)p[i] / (rho * rho);
}

/* host order -> cell order and back, for one component; stride in elements of T lets the
 * same kernel address a field inside the record array */
template <typename T>
__global__ __launch_bounds__(SPH_BLOCK) void k_put(T* __restrict__ dst, size_t stride, const T* __restrict__ src_orig,
                                                   const uint32_t* __restrict__ id, uint32_t n) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < n && id[i] != SPH_HALO_ID) dst[(size_t)i * stride] = src_orig[id[i]];
}
template <typename T>
__global__ __launch_bounds__(SPH_BLOCK) void k_get(const T* __restrict__ src, size_t stride, T* __restrict__ dst_orig,
                                                   const uint32_t* __restrict__ id, uint32_t n) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < n && id[i] != SPH_HALO_ID) dst_orig[id[i]] = src[(size_t)i * stride];
}
template <typename T>
__global__ __launch_bounds__(SPH_BLOCK) void k_fill(T* __restrict__ dst, T v, uint32_t n) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < n) dst[i] = v;
}
__global__ __launch_bounds__(SPH_BLOCK) void k_iota(uint32_t* __restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < n) dst[i] = i;
}

/* main():457-458 -- zero f of the wall particles of one state */
__global__ __launch_bounds__(SPH_BLOCK) void k_zero_wall_f(double* f0, double* f1, double* f2, int dim,
                                                           const uint32_t* __restrict__ id, uint32_t n,
                                                           uint32_t bound_parts) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n || id[i] >= bound_parts) return;
    f0[i] = 0.0;
    f1[i] = 0.0;
    if (dim == 3) f2[i] = 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* slab decomposition: halo selection, packing, placing                                       */

/* flags[i] = 1 if owned particle i is needed by the left neighbour (side 0: x <= thr) or by
 * the right neighbour (side 1: x >= thr) */
__global__ __launch_bounds__(SPH_BLOCK) void k_flag_halo(const Rec* __restrict__ rec, uint32_t n, int side, double thr,
                                                         uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double x = rec[i].xv[0][0];
    flags[i] = side == 0 ? (x <= thr ? 1u : 0u) : (x >= thr ? 1u : 0u);
}
/* pos = exclusive scan of flags: out[pos[i]] = i for flagged i (ascending); entries beyond
 * `limit` are dropped (the host sees the count and reports the overflow) */
__global__ __launch_bounds__(SPH_BLOCK) void k_compact(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                                       uint32_t n, uint32_t limit, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < n && flags[i] && pos[i] < limit) out[pos[i]] = i;
}
__global__ __launch_bounds__(SPH_BLOCK) void k_gather_rec(const Rec* __restrict__ rec, const uint32_t* __restrict__ idx,
                                                          uint32_t m, Rec* __restrict__ out) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k < m) out[k] = rec[idx[k]];
}
__global__ __launch_bounds__(SPH_BLOCK) void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                          uint32_t m, uint32_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k < m) out[k] = src[idx[k]];
}
/* received record k goes to position pos[k] (pos == nullptr: base + k) */
__global__ __launch_bounds__(SPH_BLOCK) void k_scatter_rec(const Rec* __restrict__ in, const uint32_t* __restrict__ pos,
                                                           uint32_t base, uint32_t m, Rec* __restrict__ rec) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k < m) rec[pos ? pos[k] : base + k] = in[k];
}
/* after the halo region was sorted: where did the k-th received particle go */
__global__ __launch_bounds__(SPH_BLOCK) void k_halo_pos(const uint32_t* __restrict__ perm, uint32_t m, uint32_t base,
                                                        uint32_t* __restrict__ pos) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i < m) pos[perm[i]] = base + i;
}
/* sorted halo region: records and global ids gathered through the sort permutation */
__global__ __launch_bounds__(SPH_BLOCK) void k_reorder_halo(const uint32_t* __restrict__ perm, uint32_t m,
                                                            const Rec* __restrict__ rs, Rec* __restrict__ rd,
                                                            const uint32_t* __restrict__ gs, uint32_t* __restrict__ gd) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= m) return;
    rd[i] = rs[perm[i]];
    gd[i] = gs[perm[i]];
}
/* first sorted position whose key is >= bound[q], q = 0, 1 */
__global__ void k_lower_bounds(const uint32_t* __restrict__ keys, uint32_t n, uint32_t b0, uint32_t b1,
                               uint32_t* __restrict__ out) {
    for (int q = 0; q < 2; ++q) {
        const uint32_t b = q ? b1 : b0;
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (keys[mid] < b) lo = mid + 1; else hi = mid;
        }
        out[q] = lo;
    }
}
/* ---- migration: particles that left this slab's interval move to the neighbour ------------- */

/* everything a particle owns: state n, state n+1, the outputs of the last Forces call */
struct Mig {
    double nd[12]; /* x, v, V, f of state n */
    double pd[6];  /* f, V of state n+1 */
    Rec rec;       /* x, v, rho, m, p/rho^2 of state n+1 */
    float nfl[4];  /* rho, Rrho, p, m of state n */
    float pRrho, pp;
    uint32_t gid, pad;
};
struct MigArgs {
    double* nd[12];
    double* pd[6];
    float* nfl[4];
    float* pRrho;
    float* pp;
    Rec* rec;
    uint32_t* gid;
};
/* side 0: x < plane leaves to the left; side 1: x >= plane leaves to the right */
__global__ __launch_bounds__(SPH_BLOCK) void k_flag_leavers(const Rec* __restrict__ rec, uint32_t n, int side, double plane,
                                                            uint32_t* __restrict__ flags, uint32_t* __restrict__ drop) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double x = rec[i].xv[0][0];
    const uint32_t f = side == 0 ? (x < plane ? 1u : 0u) : (x >= plane ? 1u : 0u);
    flags[i] = f;
    if (f) drop[i] = 1u;
}
__global__ __launch_bounds__(SPH_BLOCK) void k_pack_mig(MigArgs a, const uint32_t* __restrict__ idx, uint32_t m,
                                                        Mig* __restrict__ out) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const uint32_t i = idx[k];
    Mig g;
    for (int q = 0; q < 12; ++q) g.nd[q] = a.nd[q] ? a.nd[q][i] : 0.0;
    for (int q = 0; q < 6; ++q) g.pd[q] = a.pd[q] ? a.pd[q][i] : 0.0;
    for (int q = 0; q < 4; ++q) g.nfl[q] = a.nfl[q][i];
    g.rec = a.rec[i];
    g.pRrho = a.pRrho[i];
    g.pp = a.pp[i];
    g.gid = a.gid[i];
    g.pad = 0;
    out[k] = g;
}
__global__ __launch_bounds__(SPH_BLOCK) void k_unpack_mig(MigArgs a, const Mig* __restrict__ in, uint32_t m, uint32_t base) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const uint32_t i = base + k;
    const Mig g = in[k];
    for (int q = 0; q < 12; ++q)
        if (a.nd[q]) a.nd[q][i] = g.nd[q];
    for (int q = 0; q < 6; ++q)
        if (a.pd[q]) a.pd[q][i] = g.pd[q];
    for (int q = 0; q < 4; ++q) a.nfl[q][i] = g.nfl[q];
    a.rec[i] = g.rec;
    a.pRrho[i] = g.pRrho;
    a.pp[i] = g.pp;
    a.gid[i] = g.gid;
}

/* Slab runs: what a rank contributes to the step's agreement.  red[0..4] are folded with max
 * over the ranks (max mu, max |f|, redo flag, the two travel bounds), red[5..7] with a sum
 * (the position-change sums of the first and the last sweep, the pair count). */
#define RED_NMAX 5
#define RED_NSUM 3
__global__ void k_pack_global(const DevScalars* __restrict__ S, double* __restrict__ red) {
    red[0] = S->maxmu;
    red[1] = S->maxf;
    red[2] = (double)S->violation;
    red[3] = S->disp;
    red[4] = S->disp_in;
    red[5] = S->errsum0;
    red[6] = S->errsum;
    red[7] = (double)S->hit_pairs;
}

/* the step's scalars once every rank's values are folded: the same arithmetic as the last
 * k_sweep_finalize, on global maxima and sums (WCXSPH3D.cpp:280-288, 293-298) */
__global__ void k_apply_global(DevParams P, DevScalars* S, const double* __restrict__ red, int full_step) {
    const double maxmu = red[0], maxf = red[1];
    S->maxmu = maxmu;
    S->violation = red[2] != 0.0 ? 1 : 0;
    S->disp = red[3];
    S->disp_in = red[4];
    S->errsum = red[6];
    S->logbase = log10(sqrt(red[5] / P.npts_d));
    S->hit_pairs = (unsigned long long)red[7];
    if (full_step) {
        S->maxf = maxf;
        const double dtf = sqrt(P.H / maxf);
        const double dtcv = P.H / (P.Cs + maxmu);
        S->dt_next = 0.3 * fmin(dtf, dtcv);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DensityReinit (WCXSPH3D.cpp:208-249; WCXSPH.cpp:204-243; WCSPH.cpp:203-243)                  */
/*                                                                                            */
/* The reference walks the particles in index order and overwrites rho in place, so particle i */
/* sees the new rho of every neighbour with a smaller index (Gauss-Seidel).  Here a particle is */
/* released once all its smaller-index neighbours are done: the frontier of released particles  */
/* is processed by one launch, which in turn releases the next frontier.  Same values as the    */
/* sequential walk, in as many launches as the longest dependency chain.                       */

/* The frontier of a level sits in one of three queues, used in turn: level L reads queue L % 3
 * (cnt[L % 3] entries), collects the particles it sets free in queue (L+1) % 3, and clears the
 * count of queue (L+2) % 3 for the level after.  A launch therefore knows where its frontier is
 * without waiting for a control word to be published by its predecessor. */
struct ReinitCtl {
    uint32_t cnt[3];
    uint32_t done; /* particles of the levels launched so far */
};

/* W2Kernel (WCXSPH3D.cpp:121-128) */
__device__ __forceinline__ double w2_kernel(const DevParams& P, double r) {
    const double q = r / P.H;
    if (!(q < 2)) return 0.0;
    const double t = 1 - 0.5 * q;
    const double t2 = t * t;
    return (t2 * t2) * (2 * q + 1) * P.correc;
}

/* Eigen's FullPivLU on an N x N matrix, then the solve of A*beta = e0 (WCXSPH3D.cpp:232-236):
 * complete pivoting (first largest entry in column-major order), rank by the default
 * threshold eps*N*|max pivot|.  All indexing is static so that the matrix stays in registers.
 * Returns false when the matrix is not invertible by that test. */
template <int N>
__device__ __forceinline__ bool fullpiv_solve_e0(double (&lu)[N][N], double (&beta)[N]) {
    int rowT[N], colT[N];
    int nonzero = N;
    double maxpivot = 0.0;
    bool stopped = false;
#pragma unroll
    for (int kk = 0; kk < N; ++kk) {
        rowT[kk] = kk;
        colT[kk] = kk;
        if (stopped) continue;
        int br = kk, bc = kk;
        double biggest = -1.0;
        /* First pivot of a SYMMETRIC matrix (the callers' moment matrices are, bit for bit): when
         * the largest diagonal entry exceeds every entry above the diagonal, the scan below would
         * pick the first diagonal entry of that size; ten comparisons instead of sixteen
         * compare-and-select steps.  Wave-uniform like everything here. */
        bool found = false;
        if (kk == 0) {
            double d = fabs(lu[0][0]), o = 0.0;
            int di = 0;
#pragma unroll
            for (int t = 1; t < N; ++t) {
                const double a = fabs(lu[t][t]);
                if (a > d) { d = a; di = t; }
            }
#pragma unroll
            for (int c = 1; c < N; ++c)
#pragma unroll
                for (int r = 0; r < c; ++r) o = fmax(o, fabs(lu[r][c]));
            if (__builtin_amdgcn_readfirstlane((int)(o < d)) != 0) { /* false for NaN entries too */
                found = true;
                biggest = d;
                br = di;
                bc = di;
            }
        }
        if (!found) {
#pragma unroll
            for (int c = kk; c < N; ++c)
#pragma unroll
                for (int r = kk; r < N; ++r) {
                    const double a = fabs(lu[r][c]);
                    if (a > biggest) { biggest = a; br = r; bc = c; }
                }
        }
        if (biggest == 0.0) {
            nonzero = kk;
            stopped = true;
            continue;
        }
        if (biggest > maxpivot) maxpivot = biggest;
        rowT[kk] = br;
        colT[kk] = bc;
        /* the transpositions as selects over fixed places (a branch on the pivot's row or column
         * would be merged into an indexed access, and the matrix would leave the registers).  Every
         * lane of the wave solves the same system, so whether a transposition is needed at all is
         * wave-uniform: the selects are skipped when the pivot already sits in row or column kk
         * (always so for the first pivot of a moment matrix, whose largest entry is the sum of the
         * weights). */
        if (__builtin_amdgcn_readfirstlane(br) != kk) {
#pragma unroll
            for (int r = kk + 1; r < N; ++r) {
                const bool sw = r == br;
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const double a = lu[kk][c], b = lu[r][c];
                    lu[kk][c] = sw ? b : a;
                    lu[r][c] = sw ? a : b;
                }
            }
        }
        if (__builtin_amdgcn_readfirstlane(bc) != kk) {
#pragma unroll
            for (int c = kk + 1; c < N; ++c) {
                const bool sw = c == bc;
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    const double a = lu[r][kk], b = lu[r][c];
                    lu[r][kk] = sw ? b : a;
                    lu[r][c] = sw ? a : b;
                }
            }
        }
#pragma unroll
        for (int r = kk + 1; r < N; ++r) lu[r][kk] /= lu[kk][kk];
#pragma unroll
        for (int r = kk + 1; r < N; ++r)
#pragma unroll
            for (int c = kk + 1; c < N; ++c) lu[r][c] -= lu[r][kk] * lu[kk][c];
    }
    const double thr = fabs(maxpivot) * (2.220446049250313e-16 * N);
    int rank = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < nonzero && fabs(lu[i][i]) > thr) ++rank;
    if (rank != N) return false;
    double c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = i == 0 ? 1.0 : 0.0;
#pragma unroll
    for (int kk = 0; kk < N; ++kk) /* c = P*e0: the row transpositions in order */
#pragma unroll
        for (int r = kk + 1; r < N; ++r) {
            const bool sw = r == rowT[kk];
            const double a = c[kk], b = c[r];
            c[kk] = sw ? b : a;
            c[r] = sw ? a : b;
        }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int cc = 0; cc < r; ++cc) c[r] -= lu[r][cc] * c[cc];
#pragma unroll
    for (int r = N - 1; r >= 0; --r) {
#pragma unroll
        for (int cc = r + 1; cc < N; ++cc) c[r] -= lu[r][cc] * c[cc];
        c[r] /= lu[r][r];
    }
#pragma unroll
    for (int kk = N - 1; kk >= 0; --kk) /* x = Q*c: the column transpositions undone */
#pragma unroll
        for (int r = kk + 1; r < N; ++r) {
            const bool sw = r == colT[kk];
            const double a = c[kk], b = c[r];
            c[kk] = sw ? b : a;
            c[r] = sw ? a : b;
        }
#pragma unroll
    for (int i = 0; i < N; ++i) beta[i] = c[i];
    return true;
}

/* pend[i] of a finished particle; a waiting one holds the number of unfinished neighbours
 * inside the radius with a smaller original index */
#define REINIT_DONE 0x80000000u

/* The one-launch walk (k_reinit_flow) keeps one queue per workgroup; its control words: word 0 is
 * the stop flag (1: a wave gave up; 2: a slab's partial walk has run dry), the tail of queue o sits
 * 64 bytes apart from the others, and beside it the number of queue o's entries that are finished */
#define REINIT_FLOW_TAIL(o) (16u * (1u + (o)))
#define REINIT_FLOW_DONE(o) (16u * (1u + (o)) + 1u)
#define REINIT_STOP_GAVE_UP 1u
#define REINIT_STOP_DRY 2u
#define REINIT_FLOW_WORDS(owners) (16u * (1u + (owners)))
#define REINIT_EMPTY 0xFFFFFFFFu

/* pend[i] = number of neighbours inside the radius with a smaller original index; particles
 * with none form the first frontier.  Slab runs (touches != nullptr): `id` holds global
 * indices, neighbours at positions >= P.n are the adjacent ranks' particles, and touches[i]
 * says whether particle i has any of those inside the radius. */
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_count(DevParams P, const Rec* __restrict__ rec,
                                                            const uint32_t* __restrict__ nbr, uint32_t stride,
                                                            const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                            uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                            ReinitCtl* __restrict__ ctl, uint32_t* __restrict__ touches,
                                                            uint32_t owners, uint32_t qseg, uint32_t* __restrict__ flow) {
    /* four lanes per particle, each taking every fourth candidate */
    const uint32_t i_raw = blockIdx.x * (SPH_BLOCK / 4) + (threadIdx.x >> 2), q = threadIdx.x & 3u;
    const bool active = i_raw < P.n;
    const uint32_t i = active ? i_raw : P.n - 1u;
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    const uint32_t me = id[i], c = active ? cnt[i] : 0u;
    uint32_t lower = 0, far = 0;
    for (uint32_t k = q; k < c; k += 4) {
        const uint32_t j = nbr[(size_t)i * stride + k];
        const double d2 = dist2_exact(rec[j].xv[0][0] - xi, rec[j].xv[1][0] - yi, DIM == 3 ? rec[j].xv[2][0] - zi : 0.0);
        if (d2 < P.R2) {
            if (id[j] < me) ++lower;
            if (j >= P.n) far = 1;
        }
    }
    lower += (uint32_t)quad_perm_i<QP_XOR1>((int)lower);
    lower += (uint32_t)quad_perm_i<QP_XOR2>((int)lower);
    far |= (uint32_t)quad_perm_i<QP_XOR1>((int)far);
    far |= (uint32_t)quad_perm_i<QP_XOR2>((int)far);
    if (!active || q != 0) return;
    pend[i] = lower;
    if (touches) touches[i] = far;
    if (lower != 0) return;
    if (owners) { /* one-launch walk: into the queue of the workgroup that owns particle i */
        const uint32_t o = i % owners;
        queue[(size_t)o * qseg + atomicAdd(flow + REINIT_FLOW_TAIL(o), 1u)] = i;
    } else {
        queue[atomicAdd(&ctl->cnt[0], 1u)] = i; /* the first frontier: queue 0 */
    }
}

/* Slab runs, between two local walks: what the neighbours report of their particles that this
 * rank sees (new rho, finished or not) and the recount of the waiting particles next to them. */
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_pack(const Rec* __restrict__ rec, const uint32_t* __restrict__ pend,
                                                           const uint32_t* __restrict__ idx, uint32_t m,
                                                           uint2* __restrict__ out) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const uint32_t i = idx[k];
    out[k] = make_uint2(__float_as_uint(rec[i].rho), pend[i] == REINIT_DONE ? 1u : 0u);
}
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_unpack(const uint2* __restrict__ in, const uint32_t* __restrict__ pos,
                                                             uint32_t m, Rec* __restrict__ rec, uint32_t* __restrict__ pend) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const uint32_t slot = pos[k];
    if (in[k].y && pend[slot] != REINIT_DONE) {
        rec[slot].rho = __uint_as_float(in[k].x);
        pend[slot] = REINIT_DONE;
    }
}
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_rescan(DevParams P, const Rec* __restrict__ rec,
                                                             const uint32_t* __restrict__ nbr, uint32_t stride,
                                                             const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                             uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                             ReinitCtl* __restrict__ ctl, const uint32_t* __restrict__ touches,
                                                             uint32_t qcap, uint32_t level) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= P.n || !touches[i] || pend[i] == REINIT_DONE) return;
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    const uint32_t me = id[i], c = cnt[i];
    uint32_t lower = 0;
    for (uint32_t k = 0; k < c; ++k) {
        const uint32_t j = nbr[(size_t)i * stride + k];
        const double d2 = dist2_exact(rec[j].xv[0][0] - xi, rec[j].xv[1][0] - yi, DIM == 3 ? rec[j].xv[2][0] - zi : 0.0);
        if (d2 < P.R2 && id[j] < me && pend[j] != REINIT_DONE) ++lower;
    }
    /* the local walk has run dry, so nothing else touches pend[i] now; a waiting particle's
     * count was above zero until this recount */
    pend[i] = lower;
    /* the particles set free here join the frontier of the next level to be launched */
    if (lower == 0) queue[(size_t)(level % 3u) * qcap + atomicAdd(&ctl->cnt[level % 3u], 1u)] = i;
}

/* Several quotients by one denominator.  The compiler's double-precision division is
 *   r0 = rcp(d); r1 = fma(r0, fma(-d, r0, 1), r0); r2 = fma(r1, fma(-d, r1, 1), r1);
 *   q0 = n * r2; q = fma(fma(-d, q0, n), r2, q0)
 * wrapped in scaling steps that do nothing unless an operand or the quotient is close to the ends
 * of the exponent range (and a fix-up that only matters for zero, infinite or NaN operands: a zero
 * numerator gives zero either way).  The first line depends on the denominator alone.  For callers
 * that have ruled those cases out the quotient below has the bits of n / d
 * (tools/allsum_check.hip compares the two over ten million random pairs). */
struct shared_divisor {
    double d, r;
    __device__ __forceinline__ explicit shared_divisor(double den) : d(den) {
        const double r0 = __builtin_amdgcn_rcp(den);
        const double r1 = __builtin_fma(r0, __builtin_fma(-den, r0, 1.0), r0);
        r = __builtin_fma(r1, __builtin_fma(-den, r1, 1.0), r1);
    }
    __device__ __forceinline__ double quotient(double n) const {
        const double q0 = n * r;
        return __builtin_fma(__builtin_fma(-d, q0, n), r, q0);
    }
};

/* Ten wave-wide sums at once (the upper triangle of the 3-D moment matrix): instead of ten
 * butterflies of six steps, every step folds PAIRS of values, each keeping its half of the lanes
 * -- the sum of a in the lower half, of b in the upper -- so the work halves with the lane
 * span: 13 additions and 30 moves instead of 60 and 120.  Rows of 32 and 16 lanes are exchanged
 * by the row-swap instructions, 8 and 4 lanes by DPP moves written into half of the banks, the
 * last two steps inside a quad.  The totals end up in fixed lanes and are read from there, so
 * every lane gets the same bits; the order of additions is fixed. */
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_into_banks(double keep, double from) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(from), CTRL, 0xF, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(from), CTRL, 0xF, BANKS, false);
    return __hiloint2double(hi, lo);
}
template <int W>
__device__ __forceinline__ double rows_fold_pair(double a, double b) { /* a's sum in the even blocks of W lanes, b's in the odd */
    const uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
    const uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
    if (W == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void wave_allsum10(double (&v)[10]) {
    /* 32: lanes 0-31 keep the first of a pair, lanes 32-63 the second */
    const double w0 = rows_fold_pair<32>(v[0], v[1]), w1 = rows_fold_pair<32>(v[2], v[3]), w2 = rows_fold_pair<32>(v[4], v[5]),
                 w3 = rows_fold_pair<32>(v[6], v[7]), w4 = rows_fold_pair<32>(v[8], v[9]);
    /* 16: rows of sixteen now hold, in turn, v0 v2 v1 v3 | v4 v6 v5 v7 | v8 v8 v9 v9 */
    const double x0 = rows_fold_pair<16>(w0, w1), x1 = rows_fold_pair<16>(w2, w3), x2 = rows_fold_pair<16>(w4, w4);
    /* 8: lanes 0-7 of a row keep x0, lanes 8-15 x1 (a rotation by eight inside the row) */
    const double y0 = dpp_into_banks<0x128, 0xC>(x0, x1) + dpp_into_banks<0x128, 0x3>(x1, x0);
    const double y1 = x2 + dpp_into_banks<0x128, 0xF>(x2, x2);
    /* 4: quads 0 and 2 of a row keep y0, quads 1 and 3 y1 (shifts by four lanes) */
    double z = dpp_into_banks<0x114, 0xA>(y0, y1) + dpp_into_banks<0x104, 0x5>(y1, y0);
    z += quad_perm_d<QP_XOR2>(z);
    z += quad_perm_d<QP_XOR1>(z);
    /* row r = 0..3: quad 0 holds (v0 v2 v1 v3)[r], quad 2 (v4 v6 v5 v7)[r], quads 1 and 3 (v8 v8 v9 v9)[r] */
    v[0] = lane_value(z, 0);  v[2] = lane_value(z, 16); v[1] = lane_value(z, 32); v[3] = lane_value(z, 48);
    v[4] = lane_value(z, 8);  v[6] = lane_value(z, 24); v[5] = lane_value(z, 40); v[7] = lane_value(z, 56);
    v[8] = lane_value(z, 4);  v[9] = lane_value(z, 36);
}

/* One wavefront per released particle: the lanes share out its candidates, partial sums are
 * folded with a fixed butterfly (wave_allsum: every lane ends with the same bits), every lane
 * solves the small system, lane 0 stores rho, and each lane releases the waiting neighbours among
 * its candidates. */

/* What the one-launch walk hands to reinit_one: where finished particles publish their rho
 * ({rho, tag} in one 8-byte word, tag = the number of this DensityReinit call), the stop flag
 * and how long a wave may wait. */
struct ReinitFlow {
    unsigned long long* gran;
    uint32_t* stop;
    uint32_t gen;
    uint64_t limit;
};

/* DensityReinit of one released particle by one wavefront: the lanes share out its candidates,
 * partial sums are folded with a fixed butterfly (every lane ends with the same bits), every
 * lane solves the small system, lane 0 stores rho, and each lane tells `release` about the
 * waiting neighbours among its candidates whose last dependency this was.
 *   FLOW (the one-launch walk): a particle is released when its last smaller-index neighbour
 * is BEGUN, not finished, so that its wave has the lists, records and kernel weights in hand by
 * the time the neighbours' densities arrive; it then waits for the tagged rho of each of them.
 * Every particle it waits for is in the hands of a running wave (it was begun), and of those the
 * one with the smallest index waits for nobody: the walk cannot lock up.  Returns false when
 * the wave gave up (stop flag or time limit). */
template <int DIM, bool FLOW, typename Release>
__device__ __forceinline__ bool reinit_one(const DevParams& P, Rec* __restrict__ rec, const uint32_t* __restrict__ nbr,
                                           uint32_t stride, const uint32_t* __restrict__ cnt,
                                           const uint32_t* __restrict__ id, uint32_t* __restrict__ pend, uint32_t i,
                                           uint32_t lane, Release release, const ReinitFlow fl = ReinitFlow()) {
    constexpr int N = DIM + 1;
    /* this lane's first candidate is requested together with the particle's own data, and its
     * record before the count is known (entries past the count are stale but index the arrays):
     * one dependent trip fewer per level */
    const uint32_t j_first = nbr[(size_t)i * stride + min(lane, stride - 1u)];
    const Rec ri = rec[i];
    const uint32_t me = id[i], c = cnt[i];
    /* Abar (WCXSPH3D.cpp:218-229) is symmetric: its upper triangle, accumulated per lane */
    double A[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int cc = 0; cc < N; ++cc) A[r][cc] = 0.0;
    /* one neighbour's share of Abar is W * abar * m_j / rho_j (WCXSPH3D.cpp:226-228): everything
     * but the division is formed when the neighbour's record is read -- in the one-launch walk
     * that is before its new rho has arrived */
    auto share_of = [&](const double (&e)[N], double W, double mj, double (&num)[N][N]) {
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int cc = r; cc < N; ++cc) {
                const double abar = r == 0 ? e[cc] : e[cc] * e[r];
                num[r][cc] = W * abar * mj;
            }
    };
    auto add_A = [&](const double (&num)[N][N], double rhoj) {
        /* ten quotients with one denominator: the reciprocal and its two refinement steps are formed
         * once (shared_divisor); the same bits as ten divisions while no operand needs scaling, which
         * the wave checks (densities between 1e-3 and 1e12, shares zero or above 1e-250 -- num[0][0]
         * = W*m is the largest of a row's shares, the others carry one or two offsets on top) */
        const bool plain_ok = rhoj > 1e-3 && rhoj < 1e12 && !(fabs(num[0][0]) < 1e-150 && num[0][0] != 0.0);
        if (__ballot(!plain_ok) == 0ull) {
            const shared_divisor d(rhoj);
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int cc = r; cc < N; ++cc) A[r][cc] += d.quotient(num[r][cc]);
            return;
        }
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int cc = r; cc < N; ++cc) A[r][cc] += num[r][cc] / rhoj;
    };
    if (lane == 0) { /* the particle itself (distance 0) */
        double e[N];
        e[0] = 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) e[d + 1] = 0.0;
        double num[N][N];
        share_of(e, w2_kernel(P, 0.0), P.xsph ? (double)ri.m : P.mass, num);
        add_A(num, (double)ri.rho);
    }
    /* one candidate of the list: in the radius by the search's own distance (xj - xi, as the
     * search forms it)?  then Rij = xi - xj, W(|Rij|), m_j, rho_j */
    struct Nb {
        uint32_t j, idj;
        bool in;
        double e[N], mW, rho, num[N][N];
    };
    auto fetch_j = [&](uint32_t j) {
        Nb b;
        b.j = j;
        const Rec rj = rec[b.j];
        b.idj = id[b.j];
        const double d2 = dist2_exact(rj.xv[0][0] - ri.xv[0][0], rj.xv[1][0] - ri.xv[1][0],
                                      DIM == 3 ? rj.xv[2][0] - ri.xv[2][0] : 0.0);
        b.in = d2 < P.R2;
        b.e[0] = 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) b.e[d + 1] = ri.xv[d][0] - rj.xv[d][0];
        const double r = sqrt(DIM == 3 ? b.e[1] * b.e[1] + b.e[2] * b.e[2] + b.e[N - 1] * b.e[N - 1]
                                       : b.e[1] * b.e[1] + b.e[2] * b.e[2]);
        const double W = w2_kernel(P, r), m = P.xsph ? (double)rj.m : P.mass;
        share_of(b.e, W, m, b.num);
        b.mW = m * W;
        b.rho = (double)rj.rho; /* FLOW: the old value; right for the neighbours with a larger index */
        if (FLOW) { /* have all of this in registers before any waiting starts */
#pragma unroll
            for (int r = 0; r < N; ++r)
#pragma unroll
                for (int cc = r; cc < N; ++cc) asm volatile("" ::"v"(b.num[r][cc]));
#pragma unroll
            for (int d = 1; d < N; ++d) asm volatile("" ::"v"(b.e[d]));
            asm volatile("" ::"v"(b.mW));
        }
        return b;
    };
    auto fetch = [&](uint32_t k) { return fetch_j(nbr[(size_t)i * stride + k]); };
    /* FLOW: the new rho of the in-radius neighbours with a smaller index, as soon as each is
     * published (a device-scope load goes past whatever the caches hold).  Every branch below
     * is on a value made wave-uniform by hand, and the loop has one way out. */
    auto uniform = [](bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; };
    auto settle = [&](Nb& b) -> bool {
        const bool need = b.in && b.idj < me;
        unsigned long long g = need ? __hip_atomic_load(fl.gran + b.j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        bool miss = need && (uint32_t)(g >> 32) != fl.gen;
        bool wait = uniform(__ballot(miss) != 0ull);
        bool ok = true;
        if (wait) {
            const uint64_t t0 = wall_clock64();
            uint32_t spins = 0;
            while (wait) {
                __builtin_amdgcn_s_sleep(1);
                if (miss) g = __hip_atomic_load(fl.gran + b.j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                miss = need && (uint32_t)(g >> 32) != fl.gen;
                wait = uniform(__ballot(miss) != 0ull);
                if (uniform((++spins & 63u) == 0u)) {
                    const bool stop = uniform(__hip_atomic_load(fl.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
                    const bool late = uniform(wall_clock64() - t0 > fl.limit);
                    if (late && lane == 0) __hip_atomic_store(fl.stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (stop || late) {
                        ok = false;
                        wait = false;
                    }
                }
            }
        }
        if (need) b.rho = (double)__uint_as_float((uint32_t)g);
        return uniform(ok);
    };
    /* A particle releases the neighbours that were waiting for it.  Which ones is known as soon
     * as the candidates are in, and whoever reads this particle's new rho does so after the
     * level's end (a kernel boundary, or the workgroup barrier of the one-workgroup walk) or, in
     * the one-launch walk, waits for its tag: the counters are lowered now, so that the atomics'
     * round trips and the queueing run beside the sums and the solve instead of after them.  (A
     * neighbouring rank's particle, at a position >= P.n, is released by its owner.) */
    auto lower = [&](const Nb& b) {
        if (b.in && b.idj > me && b.j < P.n && atomicSub(&pend[b.j], 1u) == 1u) release(b.j);
    };
    /* the first 64 candidates stay in registers for the second pass */
    Nb first = fetch_j(j_first);
    first.in = first.in && lane < c;
    lower(first);
    bool ok = true; /* wave-uniform; FLOW only: false once a wait has been given up */
    if (FLOW) {
        /* beyond 64 candidates: release first, then wait */
        for (uint32_t k = lane + 64; k < c; k += 64) lower(fetch(k));
        ok = settle(first);
    }
    if (first.in) add_A(first.num, first.rho);
    for (uint32_t k0 = 64; k0 < c; k0 += 64) { /* wave-uniform trip count */
        const uint32_t k = k0 + lane;
        Nb b = fetch(min(k, c - 1u));
        b.in = b.in && k < c;
        if (FLOW) {
            if (ok) ok = settle(b);
        } else {
            lower(b);
        }
        if (b.in) add_A(b.num, b.rho);
    }
    if constexpr (N == 4) {
        double t[10] = {A[0][0], A[0][1], A[0][2], A[0][3], A[1][1], A[1][2], A[1][3], A[2][2], A[2][3], A[3][3]};
        wave_allsum10(t);
        int q = 0;
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int cc = r; cc < N; ++cc) {
                A[r][cc] = t[q++];
                A[cc][r] = A[r][cc];
            }
    } else {
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int cc = r; cc < N; ++cc) {
                A[r][cc] = wave_allsum(A[r][cc]);
                A[cc][r] = A[r][cc];
            }
    }
    const double A00 = A[0][0];
    double beta[N];
    if (!fullpiv_solve_e0<N>(A, beta)) { /* Shepard fallback (WCXSPH3D.cpp:235-236) */
        beta[0] = 1 / A00;
#pragma unroll
        for (int d = 1; d < N; ++d) beta[d] = (1 / A00) * 0.0;
    }
    /* rho = sum m_j W (beta . (1, Rij)) (WCXSPH3D.cpp:239-247) */
    double rho = 0.0;
    if (lane == 0) {
        double poly = beta[0] + beta[1] * 0.0 + beta[2] * 0.0;
        if (DIM == 3) poly = poly + beta[N - 1] * 0.0;
        rho += (P.xsph ? (double)ri.m : P.mass) * w2_kernel(P, 0.0) * poly;
    }
    auto second = [&](const Nb& b) {
        if (!b.in) return;
        double poly = beta[0] + beta[1] * b.e[1] + beta[2] * b.e[2];
        if (DIM == 3) poly = poly + beta[N - 1] * b.e[N - 1];
        rho += b.mW * poly;
    };
    second(first);
    for (uint32_t k0 = 64; k0 < c; k0 += 64) {
        const uint32_t k = k0 + lane;
        Nb b = fetch(min(k, c - 1u));
        b.in = b.in && k < c;
        if (FLOW && ok) ok = settle(b); /* published long ago: no wait */
        second(b);
    }
    rho = wave_allsum(rho);
    if (lane == 0 && ok) { /* a given-up particle's numbers are not stored */
        rec[i].rho = (float)rho;
        if (FLOW) /* rho and tag in one store that goes through to memory */
            __hip_atomic_store(fl.gran + i, ((unsigned long long)fl.gen << 32) | __float_as_uint((float)rho), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        else
            pend[i] = REINIT_DONE;
    }
    return ok;
}

#define REINIT_LEVEL_THREADS 1024 /* 256 workgroups of 16 waves: a frontier of 4096 particles in one pass */
template <int DIM>
__global__ __launch_bounds__(REINIT_LEVEL_THREADS) void k_reinit_level(DevParams P, Rec* __restrict__ rec,
                                                            const uint32_t* __restrict__ nbr, uint32_t stride,
                                                            const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                            uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                            ReinitCtl* __restrict__ ctl, uint32_t qcap, uint32_t level) {
    /* The particles a workgroup sets free are collected in LDS and get their places in the next
     * queue with ONE add to its count per workgroup: a thousand adds per level to that one word
     * would queue up at the memory side, and the add leaves each wave's chain of dependent steps. */
    constexpr uint32_t ROOM = 2048;
    __shared__ uint32_t s_list[ROOM];
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t cur = level % 3u, nxt = (level + 1u) % 3u;
    const uint32_t nfront = ctl->cnt[cur];
    const uint32_t* const front = queue + (size_t)cur * qcap;
    uint32_t* const next = queue + (size_t)nxt * qcap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->cnt[(level + 2u) % 3u] = 0; /* the queue the next level fills */
        ctl->done += nfront;
    }
    const uint32_t lane = threadIdx.x & 63u;
    /* consecutive queue entries go to different workgroups, so a short frontier is spread over
     * the compute units instead of filling the first few */
    const uint32_t wave = (threadIdx.x >> 6) * gridDim.x + blockIdx.x, nwaves = gridDim.x * (REINIT_LEVEL_THREADS / 64);
    for (uint32_t q = wave; q < nfront; q += nwaves)
        reinit_one<DIM, false>(P, rec, nbr, stride, cnt, id, pend, front[q], lane, [&](uint32_t j) {
            const uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < ROOM) s_list[slot] = j;
            else next[atomicAdd(&ctl->cnt[nxt], 1u)] = j; /* no room left: straight to the queue */
        });
    __syncthreads();
    const uint32_t mine = min(s_n, ROOM);
    if (mine == 0) return; /* uniform over the workgroup */
    if (threadIdx.x == 0) s_base = atomicAdd(&ctl->cnt[nxt], mine);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < mine; t += REINIT_LEVEL_THREADS) next[s_base + t] = s_list[t];
}

/* The whole walk in ONE launch, without levels: workgroup o owns the particles i with
 * i % gridDim.x == o and works through its own queue, which holds them in the order of their
 * release.  A wave claims the next place of its queue, waits until a particle shows up there
 * and re-initialises it (reinit_one, FLOW): it lowers the counters of the waiting neighbours at
 * once -- the one whose count reaches zero goes into its owner's queue -- gathers its own sums'
 * ingredients, waits for the published rho of its smaller-index neighbours and publishes its own.
 * A hop of the dependency chain costs one hand-over of 8 bytes through memory plus the sums and
 * the solve; the lists, records, counters and queues of the next particles are dealt with
 * meanwhile by other waves.
 *   Every workgroup must be resident (the host sizes the grid by the occupancy query).  Every
 * wave leaves: when its queue's places are used up (a workgroup knows how many particles it
 * owns), when the stop flag is up, or when it has waited `limit` ticks of the 100 MHz clock
 * for one thing (it raises the flag; the host reports the walk as stalled). */
#define REINIT_FLOW_THREADS 1024
/* Slab runs walk in rounds (`partial`): a particle that waits for a neighbouring rank's particle is
 * not released, so a rank's queues do not fill up and "my places are used up" never comes.  The
 * round is over when the walk has run dry: every entry ever pushed is finished, in every queue.
 * Each queue counts its pushes (its tail) and its finished entries; a particle pushes the ones it
 * sets free BEFORE it is counted as finished itself, so while anything is in work some queue has
 * fewer finished entries than pushes.  A watcher looks over all the queues
 * twice: if every queue's two counts are equal both times and the sum of the tails has not moved
 * in between, there was an instant (the end of the first pass) at which nothing was in work -- and
 * nothing can start without something in work.  The first wave of workgroup 0 does nothing but
 * this, and raises the stop flag with REINIT_STOP_DRY. */
__device__ __forceinline__ bool reinit_flow_dry(const uint32_t* __restrict__ flow, uint32_t owners, uint32_t lane) {
    unsigned long long tails[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long sum = 0;
        bool equal = true;
        for (uint32_t o = lane; o < owners; o += 64u) {
            const uint32_t t = __hip_atomic_load(flow + REINIT_FLOW_TAIL(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t d = __hip_atomic_load(flow + REINIT_FLOW_DONE(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            equal = equal && t == d;
            sum += t;
        }
        if (__ballot(!equal) != 0ull) return false;
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) sum += __shfl_xor(sum, w);
        tails[pass] = sum;
    }
    return tails[0] == tails[1];
}

template <int DIM>
__global__ __launch_bounds__(REINIT_FLOW_THREADS) void k_reinit_flow(DevParams P, Rec* __restrict__ rec,
                                                                     const uint32_t* __restrict__ nbr, uint32_t stride,
                                                                     const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                                     uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                                     uint32_t* __restrict__ flow, uint32_t qseg, uint64_t limit,
                                                                     unsigned long long* __restrict__ gran, uint32_t gen, int partial) {
    __shared__ uint32_t s_head;
    const uint32_t o = blockIdx.x, owners = gridDim.x;
    const uint32_t mine = P.n > o ? (P.n - o + owners - 1u) / owners : 0u;
    /* a round of a slab's walk goes on where the last one ended: behind the finished entries */
    if (threadIdx.x == 0) s_head = partial ? flow[REINIT_FLOW_DONE(o)] : 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t* const myq = queue + (size_t)o * qseg;
    ReinitFlow fl;
    fl.gran = gran;
    fl.stop = flow;
    fl.gen = gen;
    fl.limit = limit;
    /* every branch is on a value made wave-uniform by hand; the loops have one way out */
    auto uniform = [](bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; };
    bool run = true;
    if (partial && o == 0 && threadIdx.x < 64u) {
        /* the round's watcher: this wave takes no particles.  It leaves when the stop flag is up:
         * raised by itself (dry -- which is also how a finished walk looks) or by a wave that gave up */
        while (run) {
            __builtin_amdgcn_s_sleep(8);
            if (uniform(reinit_flow_dry(flow, owners, lane)) && lane == 0)
                __hip_atomic_store(flow, REINIT_STOP_DRY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            run = !uniform(__hip_atomic_load(flow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
        }
        return;
    }
    while (run) {
        uint32_t h = 0;
        if (lane == 0) h = atomicAdd(&s_head, 1u);
        h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
        run = uniform(h < mine);
        if (!run) continue;
        const uint64_t t0 = wall_clock64();
        uint32_t i = REINIT_EMPTY, spins = 0;
        bool waiting = true;
        while (waiting) {
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(myq + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            waiting = uniform(i == REINIT_EMPTY);
            if (waiting) {
                __builtin_amdgcn_s_sleep(4);
                if (uniform((++spins & 15u) == 0u)) {
                    const bool stop = uniform(__hip_atomic_load(flow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
                    const bool late = uniform(wall_clock64() - t0 > limit);
                    if (late && !stop && lane == 0) __hip_atomic_store(flow, REINIT_STOP_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (stop || late) {
                        run = false;
                        waiting = false;
                    }
                }
            }
        }
        if (!run) continue;
        run = reinit_one<DIM, true>(P, rec, nbr, stride, cnt, id, pend, i, lane, [&](uint32_t j) {
            const uint32_t own = j % owners;
            const uint32_t slot = atomicAdd(flow + REINIT_FLOW_TAIL(own), 1u);
            __hip_atomic_store(queue + (size_t)own * qseg + slot, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }, fl);
        /* finished (its pushes were made at its start): only now does it stop counting as in work */
        if (partial && run && lane == 0) atomicAdd(flow + REINIT_FLOW_DONE(o), 1u);
    }
}

/* Slab runs, between two rounds of the one-launch walk.  A particle is finished when its published
 * word carries this call's tag.  The neighbours are told the new rho of the finished particles they
 * see; what they report is published here under the same tag; and the waiting particles next to a
 * neighbour's particle are recounted -- the round has run dry, so nothing else touches pend[] now,
 * and every released particle is finished -- and pushed into their owners' queues when free. */
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_pack_flow(const unsigned long long* __restrict__ gran, uint32_t gen,
                                                                const uint32_t* __restrict__ idx, uint32_t m,
                                                                uint2* __restrict__ out) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const unsigned long long g = gran[idx[k]];
    out[k] = make_uint2((uint32_t)g, (uint32_t)(g >> 32) == gen ? 1u : 0u);
}
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_unpack_flow(const uint2* __restrict__ in, const uint32_t* __restrict__ pos,
                                                                  uint32_t m, Rec* __restrict__ rec,
                                                                  unsigned long long* __restrict__ gran, uint32_t gen) {
    const uint32_t k = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (k >= m) return;
    const uint32_t slot = pos[k];
    if (in[k].y && (uint32_t)(gran[slot] >> 32) != gen) {
        rec[slot].rho = __uint_as_float(in[k].x);
        gran[slot] = ((unsigned long long)gen << 32) | in[k].x;
    }
}
template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_reinit_rescan_flow(DevParams P, const Rec* __restrict__ rec,
                                                                  const uint32_t* __restrict__ nbr, uint32_t stride,
                                                                  const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                                  uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                                  uint32_t* __restrict__ flow, const uint32_t* __restrict__ touches,
                                                                  const unsigned long long* __restrict__ gran, uint32_t gen,
                                                                  uint32_t owners, uint32_t qseg) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= P.n || !touches[i] || (uint32_t)(gran[i] >> 32) == gen) return;
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    const uint32_t me = id[i], c = cnt[i];
    uint32_t lower = 0;
    for (uint32_t k = 0; k < c; ++k) {
        const uint32_t j = nbr[(size_t)i * stride + k];
        const double d2 = dist2_exact(rec[j].xv[0][0] - xi, rec[j].xv[1][0] - yi, DIM == 3 ? rec[j].xv[2][0] - zi : 0.0);
        if (d2 < P.R2 && id[j] < me && (uint32_t)(gran[j] >> 32) != gen) ++lower;
    }
    if (lower == pend[i]) return; /* nothing new next to this one (it is still waiting) */
    pend[i] = lower;
    if (lower == 0) {
        const uint32_t o = i % owners;
        queue[(size_t)o * qseg + atomicAdd(flow + REINIT_FLOW_TAIL(o), 1u)] = i;
    }
}

/* rho of the records to a plain array (back = 0) or back from it (back = 1) */
__global__ __launch_bounds__(SPH_BLOCK) void k_rho_copy(Rec* __restrict__ rec, float* __restrict__ keep, uint32_t n, int back) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (back) rec[i].rho = keep[i];
    else keep[i] = rec[i].rho;
}

/* Small cases: the whole walk in ONE workgroup, a workgroup barrier between levels instead of
 * a launch (all waves share the compute unit's cache, so a level sees the previous level's
 * stores).  A level with few particles then costs a barrier, not a kernel launch. */
#define REINIT_SINGLE_THREADS 1024
#define REINIT_SINGLE_MAX 12288 /* particles whose counters fit the workgroup's LDS */
template <int DIM>
__global__ __launch_bounds__(REINIT_SINGLE_THREADS) void k_reinit_single(DevParams P, Rec* __restrict__ rec,
                                                                         const uint32_t* __restrict__ nbr, uint32_t stride,
                                                                         const uint32_t* __restrict__ cnt,
                                                                         const uint32_t* __restrict__ id,
                                                                         uint32_t* __restrict__ pend, uint32_t* __restrict__ queue,
                                                                         ReinitCtl* __restrict__ ctl) {
    /* the queue's tail lives in LDS while the walk runs: handing out a place is an LDS atomic,
     * not a trip to memory in every wave's chain of dependent steps */
    __shared__ uint32_t s_tail;
    /* and so do the counts of unfinished smaller-index neighbours */
    __shared__ uint32_t s_pend[REINIT_SINGLE_MAX];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = REINIT_SINGLE_THREADS / 64;
    /* queue 0 holds the first frontier and, behind it, every particle in the order of its release */
    uint32_t begin = 0, end = ctl->cnt[0];
    if (threadIdx.x == 0) s_tail = end;
    for (uint32_t t = threadIdx.x; t < P.n; t += REINIT_SINGLE_THREADS) s_pend[t] = pend[t];
    __syncthreads();
    while (begin < end) { /* uniform: every thread holds the same begin and end */
        for (uint32_t q = begin + wave; q < end; q += nwaves)
            reinit_one<DIM, false>(P, rec, nbr, stride, cnt, id, s_pend, queue[q], lane,
                            [&](uint32_t j) { queue[atomicAdd(&s_tail, 1u)] = j; });
        __threadfence_block();
        __syncthreads();
        const uint32_t now = s_tail;
        __syncthreads(); /* everyone has read the tail before the next level adds to it */
        begin = end;
        end = now;
    }
    if (threadIdx.x == 0) {
        ctl->done = begin;
        ctl->cnt[0] = s_tail;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* neighbour-set probe: CSR over original indices                                             */

template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_nbr_count(DevParams P, const Rec* __restrict__ rec,
                                                         const uint32_t* __restrict__ nbr, uint32_t stride,
                                                         const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                         uint32_t* __restrict__ counts_orig) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= P.n) return;
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    uint32_t hits = dist2_exact(0.0, 0.0, 0.0) < P.R2 ? 1u : 0u; /* the particle itself */
    const uint32_t c = cnt[i];
    for (uint32_t k = 0; k < c; ++k) {
        const uint32_t j = nbr[(size_t)i * stride + k];
        const double d2 = dist2_exact(rec[j].xv[0][0] - xi, rec[j].xv[1][0] - yi, DIM == 3 ? rec[j].xv[2][0] - zi : 0.0);
        if (d2 < P.R2) ++hits;
    }
    counts_orig[id[i]] = hits;
}

template <int DIM>
__global__ __launch_bounds__(SPH_BLOCK) void k_nbr_fill(DevParams P, const Rec* __restrict__ rec,
                                                        const uint32_t* __restrict__ nbr, uint32_t stride,
                                                        const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ id,
                                                        const uint32_t* __restrict__ colid,
                                                        const uint64_t* __restrict__ offs, uint32_t* __restrict__ cols) {
    const uint32_t i = blockIdx.x * SPH_BLOCK + threadIdx.x;
    if (i >= P.n) return;
    const double xi = rec[i].xv[0][0], yi = rec[i].xv[1][0], zi = DIM == 3 ? rec[i].xv[2][0] : 0.0;
    uint32_t* row = cols + offs[id[i]];
    uint32_t m = 0;
    auto insert = [&](uint32_t v) { /* insertion into the ascending row */
        uint32_t p = m++;
        while (p > 0 && row[p - 1] > v) {
            row[p] = row[p - 1];
            --p;
        }
        row[p] = v;
    };
    if (dist2_exact(0.0, 0.0, 0.0) < P.R2) insert(colid[i]);
    const uint32_t c = cnt[i];
    for (uint32_t k = 0; k < c; ++k) {
        const uint32_t j = nbr[(size_t)i * stride + k];
        const double d2 = dist2_exact(rec[j].xv[0][0] - xi, rec[j].xv[1][0] - yi, DIM == 3 ? rec[j].xv[2][0] - zi : 0.0);
        if (d2 < P.R2) insert(colid[j]);
    }
}

#endif
