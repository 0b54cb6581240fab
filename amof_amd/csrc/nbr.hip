// Cutoff-neighbour kernels (gfx950): coordination numbers and bond angles.
//
// CN  replaces amof.atom.get_neighborlist + the counting loop of
//     amof/cn.py:58-73 (ase.neighborlist.neighbor_list('ij', atoms, dict)).
// BAD replaces amof.atom.get_neighborlist + ase.Atoms.get_angles(mic=True) +
//     numpy.histogram as driven by amof/bad.py:70-114,154-160.
//
// Both kernels only ever look at species pairs that have a cutoff: atoms are
// sorted by species into tiles (as for the RDF) and a centre tile is compared
// with the tiles of its partner species only, staged through LDS and read by
// broadcast.  All results are integer counts (exact, order independent).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "amof_internal.h"

namespace amof {

constexpr int CN_TILE = 256;
constexpr int BAD_TILE = 64;

struct NbrArgs {
    const double *pos;
    const double *geom;
    const double *img;
    const int32_t *nimg;
    const int32_t *perm;
    const Tile *tiles;
    const int32_t *sp_first_tile;  // [S]
    const int32_t *sp_ntiles;      // [S]
    const double *cutoff;          // [S][S]
    const int4 *work;              // (set/triple index, centre tile, A, B)
    int64_t N;
    int32_t F;
    int32_t n_cells;
    int32_t frames_per_chunk;
    int32_t S;
    int32_t max_img;
    int32_t n_sets;
    // CN outputs
    unsigned long long *sums;  // [F][n_sets]
    int32_t *per_atom;         // [F][n_sets][N] or null
    // BAD
    const double *edges;       // [nb+1]
    int32_t nb;
    unsigned long long *hist;  // [T][nb]
    unsigned long long *n_angles;
    int32_t *flags;            // [0] zero-length vector, [1] neighbour overflow, [2] largest neighbour count (count pass)
    int32_t cn_max;            // > 0: histograms keyed by the centre's neighbour count (BadByCn)
    // big-list pass of bad_kernel (centres with more than AMOF_MAX_NEIGHBOURS neighbours): the unit vectors live in
    // global scratch, [workgroup][3][ncap][BAD_TILE] doubles, instead of LDS
    double *nbuf;
    int32_t ncap;
    int32_t count_only;        // 1: only find the largest neighbour count (flags[2])
    int32_t global_hist;       // 1: more angle bins than LDS holds -- count with global atomics
    double edge_step;          // > 0: edges[k] == (double)k * edge_step exactly (hist_bin recomputes them); 0: read the table
    // Transposed neighbour lists (fast BAD kernels).  When both triples B-A-B and A-B-A of a species pair are asked
    // for, only the side with FEWER centres is searched; every pair it finds is also appended to the list of its
    // partner, and the angles around the other species' centres are formed from those lists (bad_transposed_kernel)
    // instead of a second search -- in ZIF-4 every N has one Zn neighbour and no angle at all, yet searching from the
    // 2304 N cost as much as the whole Zn-centred triple.  The neighbour test is symmetric (same fixed-point
    // differences negated, same canonical arithmetic), so the lists are the ones the search would have found.
    const int32_t *tr_off;     // [T] first slot of the derived triple's centres, for a searched triple; -1: none (or null)
    const int32_t *inv_rank;   // [N] position of an atom inside its species segment
    uint32_t *tcount;          // [frames of the batch][tr_total]
    uint32_t *tlist;           // [frames of the batch][tr_total][TR_CAP] partner atom indices
    int32_t tr_total;
};

constexpr int TR_CAP = 16;     // = NBRF_NLIST: a fuller centre sends the call to the exact kernels either way

// ------------------------------------------------------------------- CN ----
template <bool ORTHO, bool EXTRA>
__global__ __launch_bounds__(CN_TILE) void cn_kernel(NbrArgs a)
{
    __shared__ double tjx[CN_TILE], tjy[CN_TILE], tjz[CN_TILE];
    __shared__ int tja[CN_TILE];
    __shared__ unsigned long long wsum[CN_TILE / 64];
    const int tid = threadIdx.x;
    const int4 w = a.work[blockIdx.x];
    const int set = w.x, A = w.z, B = w.w;
    const Tile ti = a.tiles[w.y];
    const double rc = a.cutoff[A * a.S + B];
    const int64_t ai = tid < ti.count ? a.perm[ti.start + tid] : -1;
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, a.F);
    const int tb0 = a.sp_first_tile[B], tb1 = tb0 + a.sp_ntiles[B];

    for (int f = f0; f < f1; f++) {
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        const int ne = EXTRA ? a.nimg[gi] : 0;
        const double *__restrict__ E = EXTRA ? a.img + (size_t)gi * a.max_img * 3 : nullptr;
        double xi = 0.0, yi = 0.0, zi = 0.0;
        if (ai >= 0) {
            xi = p[ai * 3 + 0];
            yi = p[ai * 3 + 1];
            zi = p[ai * 3 + 2];
        }
        int cnt = 0;
        if (rc > 0.0) {
            for (int tb = tb0; tb < tb1; tb++) {
                const Tile tj = a.tiles[tb];
                __syncthreads();
                if (tid < tj.count) {
                    int64_t aj = a.perm[tj.start + tid];
                    tjx[tid] = p[aj * 3 + 0];
                    tjy[tid] = p[aj * 3 + 1];
                    tjz[tid] = p[aj * 3 + 2];
                    tja[tid] = (int)aj;
                }
                __syncthreads();
                if (ai >= 0) {
                    for (int j = 0; j < tj.count; j++) {
                        const bool self = tja[j] == (int)ai;
                        if (self && !EXTRA) continue;
                        double dx, dy, dz;
                        pair_base<ORTHO>(g, tjx[j] - xi, tjy[j] - yi, tjz[j] - zi, dx, dy, dz);
                        if (!self && sqrt(norm2(dx, dy, dz)) < rc) cnt++;
                        if (EXTRA) {
                            for (int m = 0; m < ne; m++)
                                if (sqrt(norm2(dx + E[3 * m], dy + E[3 * m + 1], dz + E[3 * m + 2])) < rc) cnt++;
                        }
                    }
                }
            }
        }
        if (a.per_atom && ai >= 0) a.per_atom[((size_t)f * a.n_sets + set) * (size_t)a.N + ai] = cnt;
        // workgroup sum of the integer counts
        unsigned long long v = ai >= 0 ? (unsigned long long)cnt : 0ull;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            unsigned long long s = 0;
            for (int k = 0; k < CN_TILE / 64; k++) s += wsum[k];
            if (s) atomicAdd(&a.sums[(size_t)f * a.n_sets + set], s);
        }
    }
}

// ------------------------------------------------------------------ BAD ----
// numpy.histogram with explicit edges: bin k holds edges[k] <= x < edges[k+1],
// the last bin is right-closed; outside -> -1.
// (e0, en = first / last edge, inv_w = nb / (en - e0): the first guess only -- the comparisons with the edges decide)
// step > 0: the host has verified edges[k] == (double)k * step bit for bit (numpy.arange(n) * dtheta, amof/bad.py:143):
// the edges are then recomputed instead of loaded -- two dependent global loads per angle otherwise
// arccos of the angle kernels.  numpy.arccos (what ase.geometry.get_angles calls) is a different routine on different CPUs and
// the device library's acos a third one; they differ in the last place for a few percent of the arguments, which bins an
// angle that sits on a histogram edge (exact lattices) differently.  Every angle here goes through ONE published algorithm
// instead -- fdlibm's acos: (asin(t) - t) / t ~ z P(z) / Q(z), z = t^2, error < 1 ulp -- in operations that are correctly
// rounded on the device (+ - * / sqrt, no contraction: -ffp-contract=off), the same algorithm the CPU oracle evaluates.
__device__ __forceinline__ double acos_pq(double z)
{
    double p = 3.47933107596021167570e-05;
    p = 7.91534994289814532176e-04 + z * p;
    p = -4.00555345006794114027e-02 + z * p;
    p = 2.01212532134862925881e-01 + z * p;
    p = -3.25565818622400915405e-01 + z * p;
    p = 1.66666666666666657415e-01 + z * p;
    p = z * p;
    double q = 7.70381505559019352791e-02;
    q = -6.88283971605453293030e-01 + z * q;
    q = 2.02094576023350569471e+00 + z * q;
    q = -2.40339491173441421878e+00 + z * q;
    q = 1.0 + z * q;
    return p / q;
}
__device__ __forceinline__ double acos_fd(double x)
{
    constexpr double HALF_PI_HEAD = 1.57079632679489655800e+00, HALF_PI_TAIL = 6.12323399573676603587e-17;
    constexpr double PI_HEAD = 3.14159265358979311600e+00;
    const double mag = fabs(x);
    if (mag >= 1.0) return x > 0.0 ? 0.0 : PI_HEAD + 2.0 * HALF_PI_TAIL;      // (the callers clip to [-1, 1])
    if (mag < 0.5) {
        if (mag < 0x1p-57) return HALF_PI_HEAD + HALF_PI_TAIL;
        const double t = x * acos_pq(x * x);
        return HALF_PI_HEAD - (x - (HALF_PI_TAIL - t));
    }
    if (x < 0.0) {
        const double z = (1.0 + x) * 0.5, root = sqrt(z);
        const double corr = acos_pq(z) * root - HALF_PI_TAIL;
        return PI_HEAD - 2.0 * (root + corr);
    }
    const double z = (1.0 - x) * 0.5, root = sqrt(z);
    const double head = __longlong_as_double(__double_as_longlong(root) & (long long)0xffffffff00000000ull);
    const double tail = (z - head * head) / (root + head);
    const double corr = acos_pq(z) * root + tail;
    return 2.0 * (head + corr);
}

__device__ __forceinline__ int hist_bin(const double *__restrict__ edges, int nb, double x, double e0, double en, double inv_w,
                                        double step = 0.0)
{
    if (!(x >= e0) || !(x <= en)) return -1;
    int k = (int)((x - e0) * inv_w);
    k = max(0, min(k, nb - 1));
    if (step > 0.0) {
        while (k > 0 && x < (double)k * step) k--;
        while (k < nb - 1 && x >= (double)(k + 1) * step) k++;
        return k;
    }
    while (k > 0 && x < edges[k]) k--;
    while (k < nb - 1 && x >= edges[k + 1]) k++;
    return k;
}

template <bool ORTHO, bool EXTRA>
__global__ __launch_bounds__(BAD_TILE) void bad_kernel(NbrArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // unit vectors of the neighbours, [slot][lane] so that a lane's accesses
    // never conflict with its neighbours' (LDS; the big-list pass keeps them in global scratch)
    double *lds_u = reinterpret_cast<double *>(lds_raw);
    const int cap = a.nbuf ? a.ncap : AMOF_MAX_NEIGHBOURS;
    double *ux = a.nbuf ? a.nbuf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 * (size_t)cap * BAD_TILE : lds_u;
    double *uy = ux + (size_t)cap * BAD_TILE;
    double *uz = uy + (size_t)cap * BAD_TILE;
    double *tjx = lds_u + 3 * AMOF_MAX_NEIGHBOURS * BAD_TILE;
    double *tjy = tjx + BAD_TILE;
    double *tjz = tjy + BAD_TILE;
    int *tja = reinterpret_cast<int *>(tjz + BAD_TILE);
    unsigned *hist = reinterpret_cast<unsigned *>(tja + BAD_TILE);

    const int tid = threadIdx.x;
    const int4 w = a.work[blockIdx.x];
    const int trip = w.x, B = w.w;
    const Tile ti = a.tiles[w.y];
    const int sa = ti.species;
    const int64_t ai = tid < ti.count ? a.perm[ti.start + tid] : -1;
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, a.F);
    const int nb = a.nb;
    const double hb_e0 = a.edges[0], hb_en = a.edges[nb], hb_inv_w = (double)nb / (hb_en - hb_e0);   // hist_bin's first guess
    const bool direct = a.cn_max > 0 || a.global_hist;     // straight into global memory (no LDS histogram)
    for (int k = tid; k < nb && !a.global_hist; k += BAD_TILE) hist[k] = 0u;
    unsigned long long nang = 0;

    for (int f = f0; f < f1; f++) {
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        const int ne = EXTRA ? a.nimg[gi] : 0;
        const double *__restrict__ E = EXTRA ? a.img + (size_t)gi * a.max_img * 3 : nullptr;
        double xi = 0.0, yi = 0.0, zi = 0.0;
        if (ai >= 0) {
            xi = p[ai * 3 + 0];
            yi = p[ai * 3 + 1];
            zi = p[ai * 3 + 2];
        }
        int n = 0;
        for (int sb = 0; sb < a.S; sb++) {
            if (!(B < 0 || sb == B)) continue;
            const double rc = a.cutoff[sa * a.S + sb];
            if (!(rc > 0.0)) continue;
            const int tb0 = a.sp_first_tile[sb], tb1 = tb0 + a.sp_ntiles[sb];
            for (int tb = tb0; tb < tb1; tb += 1) {
                // partner tiles hold up to 256 atoms: stage them 64 at a time
                const Tile tj = a.tiles[tb];
                for (int base = 0; base < tj.count; base += BAD_TILE) {
                    const int cntj = min(BAD_TILE, tj.count - base);
                    __syncthreads();
                    if (tid < cntj) {
                        int64_t aj = a.perm[tj.start + base + tid];
                        tjx[tid] = p[aj * 3 + 0];
                        tjy[tid] = p[aj * 3 + 1];
                        tjz[tid] = p[aj * 3 + 2];
                        tja[tid] = (int)aj;
                    }
                    __syncthreads();
                    if (ai < 0) continue;
                    for (int j = 0; j < cntj; j++) {
                        const bool self = tja[j] == (int)ai;
                        if (self && !EXTRA) continue;
                        double dx, dy, dz;
                        pair_base<ORTHO>(g, tjx[j] - xi, tjy[j] - yi, tjz[j] - zi, dx, dy, dz);
                        double best2 = norm2(dx, dy, dz), bx = dx, by = dy, bz = dz;
                        int hits = (!self && sqrt(best2) < rc) ? 1 : 0;
                        if (EXTRA) {
                            for (int m = 0; m < ne; m++) {
                                double ex = dx + E[3 * m], ey = dy + E[3 * m + 1], ez = dz + E[3 * m + 2];
                                double e2 = norm2(ex, ey, ez);
                                if (sqrt(e2) < rc) hits++;
                                if (e2 < best2) { best2 = e2; bx = ex; by = ey; bz = ez; }
                            }
                        }
                        if (hits) {
                            // ase.geometry.get_angles: v /= |v| before the dot product
                            double nv = sqrt(bx * bx + by * by + bz * bz);
                            if (!(nv > 0.0)) { a.flags[0] = 1; continue; }
                            double qx = bx / nv, qy = by / nv, qz = bz / nv;
                            for (int h = 0; h < hits; h++) {
                                if (a.count_only) {
                                    n++;
                                } else if (n < cap) {
                                    ux[(size_t)n * BAD_TILE + tid] = qx;
                                    uy[(size_t)n * BAD_TILE + tid] = qy;
                                    uz[(size_t)n * BAD_TILE + tid] = qz;
                                    n++;
                                } else {
                                    a.flags[1] = 1;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (a.count_only) {
            if (n > 0) atomicMax(&a.flags[2], n);
            continue;
        }
        // every unordered pair of neighbours of this centre -> one angle
        for (int u = 0; u < n; u++) {
            const double ax = ux[(size_t)u * BAD_TILE + tid], ay = uy[(size_t)u * BAD_TILE + tid], az = uz[(size_t)u * BAD_TILE + tid];
            for (int v = u + 1; v < n; v++) {
                double dot = ax * ux[(size_t)v * BAD_TILE + tid] + ay * uy[(size_t)v * BAD_TILE + tid] + az * uz[(size_t)v * BAD_TILE + tid];
                if (dot > 1.0) dot = 1.0;
                if (dot < -1.0) dot = -1.0;
                double ang = (180.0 / M_PI) * acos_fd(dot);
                int k = hist_bin(a.edges, nb, ang, hb_e0, hb_en, hb_inv_w, a.edge_step);
                if (direct) {           // BadByCn: keyed by the number of B-neighbours of this centre
                    const size_t slot = a.cn_max > 0 ? (size_t)trip * (a.cn_max + 1) + min(n, a.cn_max) : (size_t)trip;
                    atomicAdd(&a.n_angles[slot], 1ull);
                    if (k >= 0) atomicAdd(&a.hist[slot * nb + k], 1ull);
                } else {
                    nang++;
                    if (k >= 0) atomicAdd(&hist[k], 1u);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long *H = a.hist + (size_t)trip * nb;
    for (int k = tid; k < nb && !direct; k += BAD_TILE) {
        unsigned v = hist[k];
        if (v) atomicAdd(&H[k], (unsigned long long)v);
    }
    for (int off = 32; off > 0; off >>= 1) nang += __shfl_down(nang, off, 64);
    if (tid == 0 && nang) atomicAdd(&a.n_angles[trip], nang);     // (nang stays 0 in the direct modes)
}

// --------------------------------------------------------------------------
// Fast neighbour path (fully periodic cells, cutoffs that need no extra image).
//
// Shares the RDF fast path's preparation: per frame every species segment is
// folded into the cell, stored as 32-bit fixed-point fractional coordinates and
// counting-sorted into 256 slabs along the longest cell axis, with a slab offset
// table.  A centre tile only visits the partners whose slab lies within the
// cutoff of its own slab range (a 1-D cell list: 2 rc / h of the partners), the
// pair test is an f32 distance from the wrapped integer differences, and only
// pairs within 1e-6 rc of the cutoff are re-decided by the canonical float64
// arithmetic -- so every neighbour decision equals the oracle's.
struct NbrCell {
    float sc[9];         // cell rows * 2^-32 in stored axis order (ORTHO: sc[0..2])
    float _pad;
    double gap_per_len;  // 2^32 / h_axis * (1 + 1e-6): slab-key units per Angstrom of cutoff
};

struct NbrFastArgs {
    NbrArgs a;
    const QAtom *Q;               // [nf][N] species-sorted, slab-sorted
    const uint32_t *slab_start;   // [nf][S][QSLABS+1]
    const NbrCell *cells;         // [n_cells]
    const int64_t *sp_first;      // [S+1] species segment offsets
    int32_t f_base, nf;
    float guard_rel;              // relative half-width of the "re-decide exactly" band
    float guard_abs;              // absolute part (fixed-point grid), Angstrom
    // 3-D cell-list variant (CELL kernels): Q holds every atom of a frame sorted by (species, cell), x fastest
    // (launch_cell_sort, species-major keys; record idx = species << CELL_SPECIES_SHIFT | atom), cells at least as
    // thick as the largest cutoff, >= 3 per axis; start3[nf][S * ncells + 1] = offsets
    const uint32_t *start3;
    int32_t nx, ny, nz;
};

constexpr uint32_t NBR_IDX_MASK = (1u << CELL_SPECIES_SHIFT) - 1u;

template <bool ORTHO>
__device__ __forceinline__ float nbr_fast_dist(const float *sc, uint32_t uix, uint32_t uiy, uint32_t uiz, uint4 qj)
{
    const float fx = (float)(int)(qj.x - uix), fy = (float)(int)(qj.y - uiy), fz = (float)(int)(qj.z - uiz);
    float t;
    if (ORTHO) {
        const float dx = fx * sc[0], dy = fy * sc[1], dz = fz * sc[2];
        t = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    } else {
        const float dx = fmaf(fz, sc[6], fmaf(fy, sc[3], fx * sc[0]));
        const float dy = fmaf(fz, sc[7], fmaf(fy, sc[4], fx * sc[1]));
        const float dz = fmaf(fz, sc[8], fmaf(fy, sc[5], fx * sc[2]));
        t = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
    return __builtin_amdgcn_sqrtf(t);
}

// exact decision (canonical arithmetic) for a pair in the guard band
template <bool ORTHO>
__device__ __forceinline__ bool nbr_exact(const double *__restrict__ g, const double *__restrict__ p,
                                          uint32_t idx_i, uint32_t idx_j, double rc)
{
    const double *pi = p + (size_t)idx_i * 3, *pj = p + (size_t)idx_j * 3;
    double dx, dy, dz;
    pair_base<ORTHO>(g, pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2], dx, dy, dz);
    return sqrt(norm2(dx, dy, dz)) < rc;
}

// partner index range(s), relative to species B's segment, within slab reach of a centre tile
__device__ __forceinline__ void nbr_ranges(const uint32_t *__restrict__ st /* [QSLABS+1] */, uint32_t s_first,
                                           uint32_t s_last, double rc, double gap_per_len, int nB, int &b0, int &e0,
                                           int &b1, int &e1)
{
    b0 = 0; e0 = nB; b1 = 0; e1 = 0;
    const double gd = ceil(rc * gap_per_len) + 4.0;
    if (gd >= 4294967295.0) return;
    const uint32_t G = (uint32_t)gd;
    const uint32_t wlo = s_first << 24, whi = (s_last << 24) | 0xffffffu;
    const unsigned long long span = (unsigned long long)(whi - wlo) + 2ull * G + (2ull << 24);
    if (span >= (1ull << 32)) return;
    const uint32_t slo = (wlo - G) >> 24, shi = (whi + G) >> 24;
    if (slo <= shi) {
        b0 = (int)st[slo]; e0 = (int)st[shi + 1];
    } else {            // wrapped: slabs >= slo or <= shi
        b0 = 0; e0 = (int)st[shi + 1];
        b1 = (int)st[slo]; e1 = nB;
        if (b1 < e0) { e0 = nB; b1 = e1 = 0; }
    }
}

// Cell-list neighbour search of one centre atom (lane): the 27 cells around its own, as 9 rows (dz, dy) of the
// x-run cx-1 .. cx+1 (two index ranges when the run wraps), inside partner species sb's segment of the sorted
// frame.  The pair test is the fast path's: f32 distance of the wrapped fixed-point differences, pairs inside the guard band
// re-decided by the canonical float64 arithmetic.  found(atom index) is called for every neighbour.
template <bool ORTHO, typename F>
__device__ __forceinline__ void cell_neighbours(const NbrFastArgs &fa, const QAtom *__restrict__ Qf,
                                                const uint32_t *__restrict__ st, const float *sc,
                                                const double *__restrict__ geo, const double *__restrict__ p, bool has,
                                                const QAtom &qc, uint32_t own_idx, int sb, double rc, F &&found)
{
    const int nx = fa.nx, ny = fa.ny, nz = fa.nz;
    const uint32_t *__restrict__ sts = st + (size_t)sb * ((size_t)nx * ny * nz);
    const float rcf = (float)rc;
    const float g = rcf * fa.guard_rel + fa.guard_abs;
    const float r_in = rcf - g, r_out = rcf + g;
    const int cx = (int)__umulhi(qc.ux, (unsigned)nx), cy = (int)__umulhi(qc.uy, (unsigned)ny),
              cz = (int)__umulhi(qc.uz, (unsigned)nz);
    // x-run cx-1 .. cx+1 with periodic wrap (nx >= 3: three distinct cells)
    int xa0 = cx - 1, xb0 = cx + 1, xa1 = 0, xb1 = -1;
    if (xa0 < 0) { xa0 = nx - 1; xb0 = nx - 1; xa1 = 0; xb1 = cx + 1; }
    else if (xb0 >= nx) { xb0 = nx - 1; xa1 = 0; xb1 = 0; }
    // one plane of cells (dz) at a time: its six range bounds are loaded before any of its partners is gathered
    // (dependent L2 round trips otherwise); all 18 at once cost 36 live registers and a third of the occupancy
#pragma unroll 1
    for (int dz = -1; dz <= 1; dz++) {
        int cz2 = cz + dz;
        cz2 += cz2 < 0 ? nz : 0; cz2 -= cz2 >= nz ? nz : 0;
        int lo[6], hi[6];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int cy2 = cy + k - 1;
            cy2 += cy2 < 0 ? ny : 0; cy2 -= cy2 >= ny ? ny : 0;
            const int rowbase = (cz2 * ny + cy2) * nx;
            lo[2 * k] = has ? (int)sts[rowbase + xa0] : 0;
            hi[2 * k] = has ? (int)sts[rowbase + xb0 + 1] : 0;
            lo[2 * k + 1] = has && xa1 <= xb1 ? (int)sts[rowbase + xa1] : 0;
            hi[2 * k + 1] = has && xa1 <= xb1 ? (int)sts[rowbase + xb1 + 1] : 0;
        }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            for (int j = lo[r]; j < hi[r]; j += 2) {
                // two partners per trip, both gathered before either is tested
                const uint4 q0 = *reinterpret_cast<const uint4 *>(Qf + j);
                const uint4 q1 = *reinterpret_cast<const uint4 *>(Qf + min(j + 1, hi[r] - 1));
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint4 qj = u == 0 ? q0 : q1;
                    const uint32_t idx_j = qj.w & NBR_IDX_MASK;
                    if (j + u >= hi[r] || idx_j == own_idx) continue;       // (no zero-shift self pair)
                    const float d = nbr_fast_dist<ORTHO>(sc, qc.ux, qc.uy, qc.uz, qj);
                    bool nbr = d < r_in;
                    if (!nbr && d < r_out) nbr = nbr_exact<ORTHO>(geo, p, own_idx, idx_j, rc);
                    if (nbr) found(idx_j);
                }
            }
        }
    }
}

constexpr int NBRF_TILE = 256;
constexpr int NBRF_NLIST = 16;     // neighbours per centre bad_fast_kernel keeps in LDS (more: big-list pass of the exact kernel)
constexpr int NBRF_UVCAP = 768;    // unit vectors per centre tile and frame held in LDS for the flattened angle phase
// (sized for three workgroups per CU at 3600 angle bins: 18 KB unit vectors -- the partner tile aliases them --
//  + 16 KB lists + 14 KB histogram + 3.5 KB tables = 52.8 KB)
static_assert(NBRF_NLIST <= AMOF_MAX_NEIGHBOURS, "the fast kernel's lists must not exceed the documented capacity");

template <bool ORTHO, bool CELL = false>
__global__ __launch_bounds__(NBRF_TILE) void cn_fast_kernel(NbrFastArgs fa)
{
    const NbrArgs &a = fa.a;
    __shared__ uint4 tq[NBRF_TILE];
    __shared__ unsigned long long wsum[NBRF_TILE / 64];
    const int tid = threadIdx.x;
    const int4 w = a.work[blockIdx.x];          // (set, first centre (species-relative), A, B)
    const int set = w.x, c0 = w.y, A = w.z, B = w.w;
    const double rc = a.cutoff[A * a.S + B];
    const int64_t segA = fa.sp_first[A], segB = fa.sp_first[B];
    const int nA = (int)(fa.sp_first[A + 1] - segA), nB = (int)(fa.sp_first[B + 1] - segB);
    const int cnt_c = min(NBRF_TILE, nA - c0);
    const bool has = tid < cnt_c;
    const float rcf = (float)rc;
    const float g = rcf * fa.guard_rel + fa.guard_abs;
    const float r_in = rcf - g, r_out = rcf + g;
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, fa.nf);
    for (int fl = f0; fl < f1; fl++) {
        const int f = fa.f_base + fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ geo = a.geom + (size_t)gi * GEOM_STRIDE;
        const NbrCell *__restrict__ cell = fa.cells + gi;
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        float sc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) sc[k] = cell->sc[k];
        const QAtom qc = Qf[segA + c0 + min(tid, cnt_c - 1)];
        const uint32_t own_idx = CELL ? (qc.idx & NBR_IDX_MASK) : qc.idx;
        int cnt = 0;
        if (CELL) {
            if (rc > 0.0)
                cell_neighbours<ORTHO>(fa, Qf, fa.start3 + (size_t)fl * ((size_t)a.S * fa.nx * fa.ny * fa.nz + 1), sc, geo, p,
                                       has, qc, own_idx, B, rc, [&](uint32_t) { cnt++; });
        } else if (rc > 0.0) {
            const uint32_t s_first = Qf[segA + c0].uz >> 24, s_last = Qf[segA + c0 + cnt_c - 1].uz >> 24;
            int rb[2], re[2];
            nbr_ranges(fa.slab_start + ((size_t)fl * a.S + B) * (QSLABS + 1), s_first, s_last, rc, cell->gap_per_len, nB,
                       rb[0], re[0], rb[1], re[1]);
            for (int r = 0; r < 2; r++) {
                for (int j0 = rb[r]; j0 < re[r]; j0 += NBRF_TILE) {
                    const int nj = min(NBRF_TILE, re[r] - j0);
                    __syncthreads();
                    if (tid < nj) {
                        const QAtom q = Qf[segB + j0 + tid];
                        tq[tid] = make_uint4(q.ux, q.uy, q.uz, q.idx);
                    }
                    __syncthreads();
                    if (has) {
                        for (int j = 0; j < nj; j++) {
                            const uint4 qj = tq[j];
                            if (qj.w == qc.idx) continue;                       // no zero-shift self pair
                            const float d = nbr_fast_dist<ORTHO>(sc, qc.ux, qc.uy, qc.uz, qj);
                            if (d < r_in) cnt++;
                            else if (d < r_out && nbr_exact<ORTHO>(geo, p, qc.idx, qj.w, rc)) cnt++;
                        }
                    }
                }
            }
        }
        if (a.per_atom && has) a.per_atom[((size_t)f * a.n_sets + set) * (size_t)a.N + own_idx] = cnt;
        unsigned long long v = has ? (unsigned long long)cnt : 0ull;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            unsigned long long s = 0;
            for (int k = 0; k < NBRF_TILE / 64; k++) s += wsum[k];
            if (s) atomicAdd(&a.sums[(size_t)f * a.n_sets + set], s);
        }
    }
}

// ---- whole frame in LDS ------------------------------------------------------------------------------------------------
// The cell-list kernels above read a frame three times from global memory (quantise + sort into Q, the cell-start tables,
// then ~45 divergent gathers per centre through L1) and are bound by those round trips, not by arithmetic: 0.25 ms
// (quantize_cells_kernel) + 0.31 ms (cn_fast_kernel<., true>) per 5000 frames for 576 Zn centres x 13 N candidates each.
// When the two species of a pair fit in LDS (16 B per atom + 4 B per cell and species, <= 152 KB: up to ~8000 atoms
// per pair), ONE workgroup does the whole frame: fold + quantise the positions (read once, coalesced by species
// segment), counting-sort them by cell with the counters in LDS, and search from LDS -- nothing is written back but
// the result.  Every item (species pair) has its own grid, as fine as its own cutoff and the LDS budget allow.
// Decisions are those of the gather kernels: f32 distance of the wrapped fixed-point differences, pairs inside the guard
// band re-decided by the canonical float64 arithmetic on the original positions.
constexpr int NBRW_THREADS = 1024;
constexpr int NBRW_MAX_ATOMS = NBRW_THREADS * 8;      // atoms of one item (a thread folds up to 8: frame_sort<PT>)
constexpr uint32_t NBRW_SLOT1 = 0x80000000u;          // record idx word: atom index | second species of the item

struct FrameItem {
    int32_t sa, sb;            // centre species, partner species (sa == sb: one species staged)
    int32_t nx, ny, nz;        // this item's cell grid (cells >= its cutoff, >= 3 per axis); a slab: nz = ITS layers (core + 2)
    int32_t set;               // CN: output set
    int32_t reg_ab, reg_ba;    // lists: first row of region (sa, sb) / (sb, sa); -1: direction not wanted
    // z-slabs (the streaming kernels, PT = 0): a workgroup keeps the layers zoff .. zoff + nz - 1 (mod nzg) of the pair's
    // grid of nzg layers -- centres from its core layers c0 .. c0 + cn - 1 (local), partners from all of them (the core
    // and one halo layer either side).  A whole frame is zoff = 0, nz = nzg, c0 = 0, cn = nz (layers wrap as before).
    int32_t zoff, nzg, c0, cn;
    int32_t cap;               // slab kernels: records the workgroup's LDS holds (more atoms in the slab: the call falls back)
};

struct FrameArgs {
    const FrameItem *items;
    const NbrCell *cells;      // [n_cells] cell rows * 2^-32, own axis order
    const int64_t *sp_first;   // [S+1]
    int32_t *qflag;            // raised for atoms absurdly far from the cell
    int32_t f_base, nf;
    float guard_rel, guard_abs;
    // COMPACT kernels (pairs too big for two workgroups per CU with 16-byte records): 8-byte records -- 16-bit fixed point,
    // no atom index -- and, in global scratch, (atom, rank inside its species) of every sorted position
    float guard_abs16;         // the absolute guard of 16-bit coordinates
    uint2 *sidx;               // [frames of the batch][items of the launch][sidx_stride]
    int32_t sidx_stride;
    // slab kernels: the batch quantised beforehand (quantize_kernel: axis order (0, 1, 2), every species segment counting-
    // sorted into 256 bins along z), so that a slab reads only the bins its layers can lie in
    const QAtom *Q;            // [frames of the batch][N]
    const uint32_t *zstart;    // [frames of the batch][S][QSLABS + 1] first record of every bin, relative to the species segment
};

struct FrameLds {
    unsigned char *rec;        // [atoms of the item] sorted records (uint4, COMPACT: uint2): species slot 0 first (cells x fastest), then slot 1
    uint32_t *cell_end;        // [ncell] end of every cell of the sorted species (its start = the entry before it)
    uint2 *sidx;               // COMPACT: this workgroup's piece of FrameArgs::sidx
};

// record of sorted position j as (ux, uy, uz, atom | slot flag) -- COMPACT: the 16-bit coordinates in the upper halves, w = 0
template <bool COMPACT>
__device__ __forceinline__ uint4 frame_rec(const FrameLds &L, int j)
{
    if (!COMPACT) return reinterpret_cast<const uint4 *>(L.rec)[j];
    const uint2 r = reinterpret_cast<const uint2 *>(L.rec)[j];
    return make_uint4(r.x << 16, r.x & 0xffff0000u, r.y << 16, 0u);
}

template <bool COMPACT>
__device__ __forceinline__ uint32_t frame_atom(const FrameLds &L, int j, const uint4 &q)
{
    return COMPACT ? L.sidx[j].x : (q.w & ~NBRW_SLOT1);
}

// fold, quantise and cell-sort the item's one or two species of frame f into LDS.  PT = atoms per thread (records wait
// in registers between the counting and the placement pass); the index and position loads of a thread's atoms are
// issued together -- one round trip each instead of one per atom
template <int PT, bool COMPACT>
__device__ __forceinline__ void frame_sort(const NbrArgs &a, const FrameArgs &fr, const FrameItem &it, const FrameLds &L,
                                           int f, int nA, int nB, unsigned *wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = it.nx, ny = it.ny, nz = it.nz, ncell = nx * ny * nz;
    // two species: the centres (slot 0) keep their order in positions [0, nA) -- nobody looks them up by cell -- and only the
    // partners are counting-sorted behind them, so the whole table budget buys cells for ONE species
    const int n = nA + nB, ntab = ncell;
    const double *__restrict__ g = a.geom + (size_t)(a.n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const int64_t segA = fr.sp_first[it.sa], segB = fr.sp_first[it.sb];
    int32_t atom[PT];
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int k = min(tid + i * NBRW_THREADS, n - 1);
        atom[i] = a.perm[k >= nA ? segB + (k - nA) : segA + k];
    }
    for (int c = tid; c < ntab; c += NBRW_THREADS) L.cell_end[c] = 0u;
    double px[PT], py[PT], pz[PT];
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const double *__restrict__ pp = a.pos + ((size_t)f * (size_t)a.N + (size_t)atom[i]) * 3;
        px[i] = pp[0]; py[i] = pp[1]; pz[i] = pp[2];
    }
    __syncthreads();
    uint4 rec[PT];
    uint32_t key[PT];
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int k = tid + i * NBRW_THREADS;
        uint32_t u[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {       // (quantize_atom's arithmetic on the positions already loaded)
            double sf = fma(pz[i], g[15 + c], fma(py[i], g[12 + c], px[i] * g[9 + c]));
            if (!(fabs(sf) < 1.0e4)) *fr.qflag = 1;
            sf = sf - floor(sf);
            const double t = sf * 4294967296.0;
            u[c] = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
            if (COMPACT) u[c] &= 0xffff0000u;       // (the cell of an atom follows from the coordinate its record keeps)
        }
        const bool second = k >= nA;
        rec[i] = make_uint4(u[0], u[1], u[2], (uint32_t)atom[i] | (second ? NBRW_SLOT1 : 0u));
        key[i] = (__umulhi(u[2], (unsigned)nz) * (unsigned)ny + __umulhi(u[1], (unsigned)ny)) * (unsigned)nx + __umulhi(u[0], (unsigned)nx);
        if (k < n && (second || nB == 0)) atomicAdd(&L.cell_end[key[i]], 1u);
    }
    __syncthreads();
    // exclusive scan of the counters, in place: one contiguous chunk per thread, chunk totals scanned by waves
    const int chunk = (ntab + NBRW_THREADS - 1) / NBRW_THREADS;
    const int c0 = min(tid * chunk, ntab), c1 = min(c0 + chunk, ntab);
    unsigned sum = 0;
    for (int c = c0; c < c1; c++) sum += L.cell_end[c];
    unsigned incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum + (nB > 0 ? (unsigned)nA : 0u);      // (sorted positions start behind the centres)
    for (int q = 0; q < wave; q++) run += wsum[q];
    for (int c = c0; c < c1; c++) {
        const unsigned v = L.cell_end[c];
        L.cell_end[c] = run;                // (cursor of the placement pass; ends as the cell's end)
        run += v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT; i++) {
        const int k = tid + i * NBRW_THREADS;
        if (k < n) {
            const unsigned slot = (k >= nA || nB == 0) ? atomicAdd(&L.cell_end[key[i]], 1u) : (unsigned)k;
            if (COMPACT) {
                reinterpret_cast<uint2 *>(L.rec)[slot] = make_uint2((rec[i].x >> 16) | rec[i].y, rec[i].z >> 16);
                L.sidx[slot] = make_uint2(rec[i].w & ~NBRW_SLOT1, (uint32_t)(k >= nA ? k - nA : k));
            } else {
                reinterpret_cast<uint4 *>(L.rec)[slot] = rec[i];
            }
        }
    }
    __syncthreads();
}

// ---- z-slabs (round 4): pairs with more atoms than one workgroup's LDS (or its registers: frame_sort keeps a thread's atoms
// in registers between its two passes) holds.  The batch is quantised beforehand into records sorted by species and by 256
// bins along z (quantize_kernel, the RDF path's: positions read once per call instead of once per pair); a workgroup reads
// the bins its layers can lie in -- twice: count, then place -- and keeps the partners of its core layers and of one halo
// layer either side, and the centres of its core layers.  Two species: the centres are compacted (in arrival order) to
// positions [0, ncen), the partners cell-sorted behind them.  One species: everything is cell-sorted and the centres are the
// contiguous run of the core layers.
// Returns false when the slab holds more atoms than it.cap (then *fr.qflag is raised: the gather kernels take the call).
constexpr int NBRS_RT = 4;      // records per thread and round
__device__ __forceinline__ bool frame_sort_slab(const NbrArgs &a, const FrameArgs &fr, const FrameItem &it, const FrameLds &L,
                                                int f, int nB, unsigned *wsum, unsigned *ctl /* LDS [4] */,
                                                int &ncen, int &cbeg, int &first, int &total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = it.nx, ny = it.ny, nzl = it.nz, ntab = nx * ny * nzl;
    const bool two = nB > 0;
    const uint4 *__restrict__ Qf = reinterpret_cast<const uint4 *>(fr.Q + (size_t)(f - fr.f_base) * (size_t)a.N);
    const uint32_t *__restrict__ zs = fr.zstart + (size_t)(f - fr.f_base) * (size_t)a.S * (QSLABS + 1);
    for (int c = tid; c < ntab; c += NBRW_THREADS) L.cell_end[c] = 0u;
    if (tid < 4) ctl[tid] = 0u;
    __syncthreads();
    // The records of species sp whose layer can be one of l0 .. l0 + ln - 1 (mod nzg): up to two runs of its z-sorted
    // segment.  fn(record, is a partner of this slab, is a centre of this slab, its cell) is called by all lanes for
    // every record of the runs, so that fn may aggregate over the wave.
    auto walk = [&](int sp, int l0, int ln, bool centres, auto &&fn) {
        const uint32_t *__restrict__ st = zs + (size_t)sp * (QSLABS + 1);
        const int seg = (int)fr.sp_first[sp];
        l0 -= l0 >= it.nzg ? it.nzg : 0;
        const int b0 = (l0 * QSLABS) / it.nzg, b1 = ((l0 + ln) * QSLABS + it.nzg - 1) / it.nzg;
        int r0, e0, r1 = 0, e1 = 0;
        if (b1 - b0 >= QSLABS) { r0 = 0; e0 = (int)st[QSLABS]; }
        else if (b1 <= QSLABS) { r0 = (int)st[b0]; e0 = (int)st[b1]; }
        else { r0 = (int)st[b0]; e0 = (int)st[QSLABS]; r1 = 0; e1 = (int)st[b1 - QSLABS]; }
        const int len0 = e0 - r0, M = len0 + e1 - r1;
        for (int base = 0; base < M; base += NBRS_RT * NBRW_THREADS) {
            uint4 q[NBRS_RT];
#pragma unroll
            for (int i = 0; i < NBRS_RT; i++) {
                const int k = min(base + tid + i * NBRW_THREADS, M - 1);
                q[i] = Qf[seg + (k < len0 ? r0 + k : r1 + (k - len0))];
            }
#pragma unroll
            for (int i = 0; i < NBRS_RT; i++) {
                const int k = base + tid + i * NBRW_THREADS;
                int lz = (int)__umulhi(q[i].z, (unsigned)it.nzg) - it.zoff;
                lz += lz < 0 ? it.nzg : 0;
                const bool partner = !centres && k < M && lz < nzl;
                const bool centre = centres && k < M && (unsigned)(lz - it.c0) < (unsigned)it.cn;
                const uint32_t key = ((unsigned)min(lz, nzl - 1) * (unsigned)ny + __umulhi(q[i].y, (unsigned)ny)) * (unsigned)nx + __umulhi(q[i].x, (unsigned)nx);
                fn(make_uint4(q[i].x, q[i].y, q[i].z, q[i].w | ((two && !centres) ? NBRW_SLOT1 : 0u)), partner, centre, key);
            }
        }
    };
    auto walk_all = [&](auto &&fn) {
        if (two) walk(it.sa, it.zoff + it.c0, it.cn, true, fn);
        walk(it.sb, it.zoff, nzl, false, fn);
    };
    walk_all([&](const uint4 &, bool partner, bool centre, uint32_t key) {
        if (partner) atomicAdd(&L.cell_end[key], 1u);
        if (two) {
            const unsigned long long m = __ballot(centre);
            if (lane == 0 && m) atomicAdd(&ctl[0], (unsigned)__popcll(m));
        }
    });
    __syncthreads();
    ncen = two ? (int)ctl[0] : 0;
    first = ncen;
    // exclusive scan of the counters, as in frame_sort; the thread that holds the table's end publishes the total
    const int chunk = (ntab + NBRW_THREADS - 1) / NBRW_THREADS;
    const int c0 = min(tid * chunk, ntab), c1 = min(c0 + chunk, ntab);
    unsigned sum = 0;
    for (int c = c0; c < c1; c++) sum += L.cell_end[c];
    unsigned incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum + (unsigned)ncen;
    for (int q = 0; q < wave; q++) run += wsum[q];
    for (int c = c0; c < c1; c++) {
        const unsigned v = L.cell_end[c];
        L.cell_end[c] = run;
        run += v;
    }
    if (c0 < c1 && c1 == ntab) ctl[1] = run;
    __syncthreads();
    total = (int)ctl[1];
    if (total > it.cap) {
        if (tid == 0) *fr.qflag = 1;
        return false;
    }
    walk_all([&](const uint4 &rec, bool partner, bool centre, uint32_t key) {
        if (partner) reinterpret_cast<uint4 *>(L.rec)[atomicAdd(&L.cell_end[key], 1u)] = rec;
        if (two) {
            const unsigned long long m = __ballot(centre);
            if (m) {
                unsigned b = 0;
                if (lane == 0) b = atomicAdd(&ctl[2], (unsigned)__popcll(m));
                b = __shfl(b, 0, 64) + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (centre) reinterpret_cast<uint4 *>(L.rec)[b] = rec;
            }
        }
    });
    __syncthreads();
    if (two) {
        cbeg = 0;
    } else {    // one species: the centres are the sorted run of the core layers
        const int lo = it.c0 * ny * nx, hi = (it.c0 + it.cn) * ny * nx;
        cbeg = lo > 0 ? (int)L.cell_end[lo - 1] : 0;
        ncen = (int)L.cell_end[hi - 1] - cbeg;
    }
    return true;
}

// One task = one centre x one of the 9 rows (dz, dy) of cells around it: its x-run cx-1 .. cx+1 (two index ranges when
// the run wraps) among the partners of species slot `slot` (0 / 1) of the sorted frame.  Nine lanes share a centre, so a
// wave's trip count is the fullest ROW of its lanes, not the fullest neighbourhood (per-lane loops over all 27 cells ran
// at a third of the issue rate: 18 us of a 31 us frame).  visit(is a neighbour, sorted position of the partner) is called
// for EVERY candidate by all lanes still in the loop, so that a caller may aggregate over the wave.
// c = the centre's own sorted position; one_species: partners and centres are the same atoms (the zero-shift self pair is skipped).
template <bool ORTHO, bool COMPACT, typename F>
__device__ __forceinline__ void frame_row_neighbours(const FrameArgs &fr, const FrameItem &it, const FrameLds &L, int first,
                                                     const float *sc, const double *__restrict__ geo,
                                                     const double *__restrict__ p, int c, bool one_species, const uint4 qc, int r9, double rc,
                                                     F &&visit)
{
    const int nx = it.nx, ny = it.ny, nz = it.nz;
    const float rcf = (float)rc;
    const float gd = rcf * fr.guard_rel + (COMPACT ? fr.guard_abs16 : fr.guard_abs);
    const float r_in = rcf - gd, r_out = rcf + gd;
    const int cx = (int)__umulhi(qc.x, (unsigned)nx), cy = (int)__umulhi(qc.y, (unsigned)ny);
    int cz = (int)__umulhi(qc.z, (unsigned)it.nzg) - it.zoff;      // (the layer inside this workgroup's slab; a whole frame: zoff = 0)
    cz += cz < 0 ? it.nzg : 0;
    const int dz = r9 / 3 - 1, dy = r9 - 3 * (r9 / 3) - 1;
    int cz2 = cz + dz, cy2 = cy + dy;
    cz2 += cz2 < 0 ? nz : 0; cz2 -= cz2 >= nz ? nz : 0;
    cy2 += cy2 < 0 ? ny : 0; cy2 -= cy2 >= ny ? ny : 0;
    const int row = (cz2 * ny + cy2) * nx;          // (first = sorted position of the first partner: the start of cell 0)
    // cells xa .. xb of the row, plus the wrapped piece (nx >= 3: three distinct cells)
    int xa = cx - 1, xb = cx + 1, wa = 0, wb = -1;
    if (xa < 0) { xa = 0; wa = nx - 1; wb = nx - 1; }
    else if (xb >= nx) { xb = nx - 1; wa = 0; wb = 0; }
    const int lo0 = row + xa > 0 ? (int)L.cell_end[row + xa - 1] : first, hi0 = (int)L.cell_end[row + xb];
    int lo1 = 0, hi1 = 0;
    if (wa <= wb) { lo1 = row + wa > 0 ? (int)L.cell_end[row + wa - 1] : first; hi1 = (int)L.cell_end[row + wb]; }
    const int len0 = hi0 - lo0, total = len0 + hi1 - lo1;
    for (int q = 0; q < total; q++) {
        const int j = q < len0 ? lo0 + q : lo1 + (q - len0);
        const uint4 qj = frame_rec<COMPACT>(L, j);
        const float d = nbr_fast_dist<ORTHO>(sc, qc.x, qc.y, qc.z, qj);
        bool nbr = d < r_in;
        if (!nbr && d < r_out) nbr = nbr_exact<ORTHO>(geo, p, frame_atom<COMPACT>(L, c, qc), frame_atom<COMPACT>(L, j, qj), rc);
        visit(nbr && !(one_species && j == c), j);              // (no zero-shift self pair)
    }
}

// PT = 4: up to 4096 atoms per item, two workgroups per CU (64 VGPRs); PT = 8: up to 8192 -- two per CU with COMPACT records when
// the pair then fits half a CU's LDS, one otherwise
template <bool ORTHO, int PT, bool COMPACT>
__global__ __launch_bounds__(NBRW_THREADS, (PT == 8 && !COMPACT) ? 4 : 8) void cn_frame_kernel(NbrArgs a, FrameArgs fr)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ unsigned wsum[NBRW_THREADS / 64];
    __shared__ unsigned long long wsum64[NBRW_THREADS / 64];
    FrameLds L;
    const int tid = threadIdx.x;
    const FrameItem it = fr.items[blockIdx.x];
    const int f = fr.f_base + (int)blockIdx.y;
    const int nA = (int)(fr.sp_first[it.sa + 1] - fr.sp_first[it.sa]);
    const int nB = it.sa == it.sb ? 0 : (int)(fr.sp_first[it.sb + 1] - fr.sp_first[it.sb]);
    constexpr bool SLAB = PT == 0;
    static_assert(!(SLAB && COMPACT), "slabs keep 16-byte records");
    L.rec = lds_raw;                                                    // (every item lays LDS out for its own atom count)
    L.cell_end = reinterpret_cast<uint32_t *>(lds_raw + (size_t)(SLAB ? it.cap : nA + nB) * (COMPACT ? sizeof(uint2) : sizeof(uint4)));
    L.sidx = COMPACT ? fr.sidx + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)fr.sidx_stride : nullptr;
    const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
    int32_t *__restrict__ pa = a.per_atom ? a.per_atom + ((size_t)f * a.n_sets + it.set) * (size_t)a.N : nullptr;
    int ncen = nA, cbeg = 0, first = nB > 0 ? nA : 0;
    if constexpr (SLAB) {
        __shared__ unsigned ctl[4];
        int total;
        if (!frame_sort_slab(a, fr, it, L, f, nB, wsum, ctl, ncen, cbeg, first, total)) return;
        if (pa) {   // the centres of THIS slab start at zero (a centre belongs to one slab: nobody else adds to it)
            for (int c = tid; c < ncen; c += NBRW_THREADS) pa[reinterpret_cast<const uint4 *>(L.rec)[cbeg + c].w & ~NBRW_SLOT1] = 0;
            __syncthreads();
        }
    } else {
        if (pa)     // every centre starts at zero (the barriers of the sort order these stores before the atomics below)
            for (int c = tid; c < nA; c += NBRW_THREADS) pa[a.perm[fr.sp_first[it.sa] + c]] = 0;
        frame_sort<PT, COMPACT>(a, fr, it, L, f, nA, nB, wsum);
    }
    const int gi = a.n_cells == 1 ? 0 : f;
    const double *__restrict__ geo = a.geom + (size_t)gi * GEOM_STRIDE;
    float sc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sc[k] = fr.cells[gi].sc[k];
    const double rc = a.cutoff[it.sa * a.S + it.sb];
    unsigned long long sum = 0;
    const int tasks = ncen * 9;
    for (int t = tid; t < tasks; t += NBRW_THREADS) {
        const int c = cbeg + t / 9, r9 = t % 9;
        const uint4 qc = frame_rec<COMPACT>(L, c);
        int cnt = 0;
        frame_row_neighbours<ORTHO, COMPACT>(fr, it, L, first, sc, geo, p, c, nB == 0, qc, r9, rc,
                                             [&](bool nbr, int) { cnt += nbr ? 1 : 0; });
        if (pa && cnt) atomicAdd(&pa[frame_atom<COMPACT>(L, c, qc)], cnt);
        sum += (unsigned long long)cnt;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((tid & 63) == 0) wsum64[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long s2 = 0;
        for (int k = 0; k < NBRW_THREADS / 64; k++) s2 += wsum64[k];
        if (s2) atomicAdd(&a.sums[(size_t)f * a.n_sets + it.set], s2);
    }
}

// ase.geometry.get_angles on two canonical minimum-image vectors (normalise, dot, clip, acos)
__device__ __forceinline__ bool unit_vec(double x, double y, double z, double &ux, double &uy, double &uz)
{
    const double nv = sqrt(x * x + y * y + z * z);
    if (!(nv > 0.0)) return false;
    ux = x / nv; uy = y / nv; uz = z / nv;
    return true;
}

// One lane = one centre atom for the neighbour search (as in cn_fast_kernel); the angles are then formed by the
// whole workgroup over FLATTENED work lists, because per-lane loops leave most lanes idle (in ZIF-4 every N has one
// Zn neighbour and no angle, every Zn has four N and six angles) and chain dependent gathers (the old per-lane loop:
// ~9 global round trips per frame, 8.1 ms per 5000 frames; this form: profiles/r02):
//   2a  entries (centre c, neighbour slot u), all centres of the tile: gather the two positions once, canonical
//       minimum-image vector, unit vector -> LDS;
//   2b  every entry forms its angles with the later entries of the same centre from LDS -- no global loads.
// Tiles whose centres hold more neighbours than the LDS table (NBRF_UVCAP entries) go through it in groups of centres.
// Same arithmetic per angle as before (ase get_angles order), so the counts still equal the oracle's bit for bit.
template <bool ORTHO, bool CELL = false>
__global__ __launch_bounds__(NBRF_TILE) void bad_fast_kernel(NbrFastArgs fa)
{
    const NbrArgs &a = fa.a;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    uint4 *tq = reinterpret_cast<uint4 *>(lds_raw);                       // [NBRF_TILE] (search phase only: aliases uv)
    double *uvx = reinterpret_cast<double *>(lds_raw);                    // [NBRF_UVCAP] x 3 (angle phase only)
    static_assert(3 * NBRF_UVCAP * sizeof(double) >= NBRF_TILE * sizeof(uint4), "partner tile must fit in the uv table");
    double *uvy = uvx + NBRF_UVCAP, *uvz = uvy + NBRF_UVCAP;
    uint32_t *nlist = reinterpret_cast<uint32_t *>(uvz + NBRF_UVCAP);     // [NBRF_NLIST][NBRF_TILE]
    uint32_t *s_cidx = nlist + NBRF_NLIST * NBRF_TILE;                    // [NBRF_TILE] atom index of every centre
    int *pref = reinterpret_cast<int *>(s_cidx + NBRF_TILE);              // [NBRF_TILE + 1] entries before centre c
    unsigned short *ec = reinterpret_cast<unsigned short *>(pref + NBRF_TILE + 4);   // [NBRF_UVCAP] centre of entry e
    unsigned *hist = reinterpret_cast<unsigned *>(ec + NBRF_UVCAP);       // [nb]
    __shared__ int s_wtot[NBRF_TILE / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int4 w = a.work[blockIdx.x];          // (triple, first centre (species-relative), centre species, B)
    const int trip = w.x, c0 = w.y, sa = w.z, B = w.w;
    const int troff = a.tr_off ? a.tr_off[trip] : -1;
    const int64_t segA = fa.sp_first[sa];
    const int nA = (int)(fa.sp_first[sa + 1] - segA);
    const int cnt_c = min(NBRF_TILE, nA - c0);
    const bool has = tid < cnt_c;
    const int nb = a.nb;
    const double hb_e0 = a.edges[0], hb_en = a.edges[nb], hb_inv_w = (double)nb / (hb_en - hb_e0);   // hist_bin's first guess
    const bool direct = a.cn_max > 0 || a.global_hist;     // straight into global memory (no LDS histogram)
    for (int k = tid; k < nb && !a.global_hist; k += NBRF_TILE) hist[k] = 0u;
    unsigned long long nang = 0;
    // one angle between two unit vectors -> histogram
    auto count_angle = [&](double ax, double ay, double az, double bx, double by, double bz, int n_centre) {
        double dot = ax * bx + ay * by + az * bz;
        if (dot > 1.0) dot = 1.0;
        if (dot < -1.0) dot = -1.0;
        const double ang = (180.0 / M_PI) * acos_fd(dot);
        const int k = hist_bin(a.edges, nb, ang, hb_e0, hb_en, hb_inv_w, a.edge_step);
        if (direct) {           // BadByCn: keyed by the number of B-neighbours of this centre
            const size_t slot = a.cn_max > 0 ? (size_t)trip * (a.cn_max + 1) + min(n_centre, a.cn_max) : (size_t)trip;
            atomicAdd(&a.n_angles[slot], 1ull);
            if (k >= 0) atomicAdd(&a.hist[slot * nb + k], 1ull);
        } else {
            nang++;
            if (k >= 0) atomicAdd(&hist[k], 1u);
        }
    };
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, fa.nf);
    for (int fl = f0; fl < f1; fl++) {
        const int f = fa.f_base + fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ geo = a.geom + (size_t)gi * GEOM_STRIDE;
        const NbrCell *__restrict__ cell = fa.cells + gi;
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        float sc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) sc[k] = cell->sc[k];
        const QAtom qc = Qf[segA + c0 + min(tid, cnt_c - 1)];
        const uint32_t own_idx = CELL ? (qc.idx & NBR_IDX_MASK) : qc.idx;
        const uint32_t s_first = CELL ? 0u : Qf[segA + c0].uz >> 24, s_last = CELL ? 0u : Qf[segA + c0 + cnt_c - 1].uz >> 24;
        int n = 0;
        auto found = [&](uint32_t idx_j) {
            if (n < NBRF_NLIST) nlist[n * NBRF_TILE + tid] = idx_j;
            else a.flags[1] = 1;
            n++;
        };
        for (int sb = 0; sb < a.S; sb++) {
            if (!(B < 0 || sb == B)) continue;
            const double rc = a.cutoff[sa * a.S + sb];
            if (!(rc > 0.0)) continue;
            if (CELL) {
                cell_neighbours<ORTHO>(fa, Qf, fa.start3 + (size_t)fl * ((size_t)a.S * fa.nx * fa.ny * fa.nz + 1), sc, geo, p,
                                       has, qc, own_idx, sb, rc, found);
                continue;
            }
            const int64_t segB = fa.sp_first[sb];
            const int nB = (int)(fa.sp_first[sb + 1] - segB);
            const float rcf = (float)rc;
            const float g = rcf * fa.guard_rel + fa.guard_abs;
            const float r_in = rcf - g, r_out = rcf + g;
            int rb[2], re[2];
            nbr_ranges(fa.slab_start + ((size_t)fl * a.S + sb) * (QSLABS + 1), s_first, s_last, rc, cell->gap_per_len,
                       nB, rb[0], re[0], rb[1], re[1]);
            for (int r = 0; r < 2; r++) {
                for (int j0 = rb[r]; j0 < re[r]; j0 += NBRF_TILE) {
                    const int nj = min(NBRF_TILE, re[r] - j0);
                    __syncthreads();
                    if (tid < nj) {
                        const QAtom q = Qf[segB + j0 + tid];
                        tq[tid] = make_uint4(q.ux, q.uy, q.uz, q.idx);
                    }
                    __syncthreads();
                    if (!has) continue;
                    for (int j = 0; j < nj; j++) {
                        const uint4 qj = tq[j];
                        if (qj.w == qc.idx) continue;
                        const float d = nbr_fast_dist<ORTHO>(sc, qc.ux, qc.uy, qc.uz, qj);
                        bool nbr = d < r_in;
                        if (!nbr && d < r_out) nbr = nbr_exact<ORTHO>(geo, p, qc.idx, qj.w, rc);
                        if (nbr) found(qj.w);
                    }
                }
            }
        }
        n = has ? min(n, NBRF_NLIST) : 0;
        if (troff >= 0) {
            // the same pairs, seen from the partners: appended to their transposed lists AFTER the search (the returning
            // atomics of a lane are independent of each other here; inside the search loop each one stalled the lane)
            for (int u = 0; u < n; u++) {
                const uint32_t idx_j = nlist[u * NBRF_TILE + tid];
                const size_t slot = (size_t)fl * a.tr_total + (size_t)(troff + a.inv_rank[idx_j]);
                const uint32_t c = atomicAdd(&a.tcount[slot], 1u);
                if (c < (uint32_t)TR_CAP) a.tlist[slot * TR_CAP + c] = own_idx;
            }
        }
        // a centre with a single neighbour -- every N of ZIF-4 -- forms no angle: only centres with >= 2 enter
        const int n_ent = n >= 2 ? n : 0;
        // (barrier: the previous frame's angle phase has finished with pref / uv / ec)  no angle in this tile and
        // frame -- e.g. a tile of N centres -- : next frame
        if (!__syncthreads_or(n_ent > 0)) continue;
        // entries before each centre: exclusive scan of n_ent over the workgroup
        int incl = n_ent;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wtot[wave] = incl;
        s_cidx[tid] = own_idx;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wave; q++) before += s_wtot[q];
        pref[tid] = before + incl - n_ent;
        if (tid == NBRF_TILE - 1) pref[NBRF_TILE] = before + incl;
        __syncthreads();
        // centres in groups whose entries fit the LDS table (one group unless the tile holds > NBRF_UVCAP neighbours;
        // a centre has at most NBRF_NLIST <= NBRF_UVCAP entries, so every group makes progress)
        for (int c_begin = 0; c_begin < NBRF_TILE;) {
            const int base = pref[c_begin];
            int lo_c = c_begin, hi_c = NBRF_TILE + 1;     // largest c_end in (c_begin, TILE] with pref[c_end] - base <= UVCAP
            while (hi_c - lo_c > 1) {
                const int mid = (lo_c + hi_c) >> 1;
                if (pref[mid] - base <= NBRF_UVCAP) lo_c = mid;
                else hi_c = mid;
            }
            const int c_end = max(lo_c, c_begin + 1);
            const int total = pref[c_end] - base;
            // 2a: one unit vector per (centre, neighbour slot)
            for (int e = tid; e < total; e += NBRF_TILE) {
                int lo = c_begin, hi = c_end;    // largest c in [c_begin, c_end) with pref[c] <= base + e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (pref[mid] <= base + e) lo = mid;
                    else hi = mid;
                }
                const int c = lo, u = base + e - pref[c];
                const double *pn = p + (size_t)nlist[u * NBRF_TILE + c] * 3, *pc = p + (size_t)s_cidx[c] * 3;
                double vx, vy, vz, ax = 0.0, ay = 0.0, az = 0.0;
                pair_base<ORTHO>(geo, pn[0] - pc[0], pn[1] - pc[1], pn[2] - pc[2], vx, vy, vz);
                if (!unit_vec(vx, vy, vz, ax, ay, az)) a.flags[0] = 1;     // (the call fails: results are discarded)
                uvx[e] = ax; uvy[e] = ay; uvz[e] = az;
                ec[e] = (unsigned short)c;
            }
            __syncthreads();
            // 2b: entry e = (c, u) with every later entry (c, v > u) of the same centre
            for (int e = tid; e < total; e += NBRF_TILE) {
                const int c = ec[e], e_end = pref[c + 1] - base, n_c = pref[c + 1] - pref[c];
                const double ax = uvx[e], ay = uvy[e], az = uvz[e];
                for (int ev = e + 1; ev < e_end; ev++) count_angle(ax, ay, az, uvx[ev], uvy[ev], uvz[ev], n_c);
            }
            c_begin = c_end;
            if (c_begin < NBRF_TILE) __syncthreads();      // the next group overwrites the table
        }
    }
    __syncthreads();
    unsigned long long *H = a.hist + (size_t)trip * nb;
    for (int k = tid; k < nb && !direct; k += NBRF_TILE) {
        unsigned v = hist[k];
        if (v) atomicAdd(&H[k], (unsigned long long)v);
    }
    for (int off = 32; off > 0; off >>= 1) nang += __shfl_down(nang, off, 64);
    if ((tid & 63) == 0 && nang) atomicAdd(&a.n_angles[trip], nang);
}

// Angles around the centres whose neighbour lists were written by the search from the OTHER side (see NbrArgs).  One
// lane per centre, frames in chunks per workgroup (LDS histogram, one flush); a centre with fewer than two partners --
// every N of an intact ZIF-4 -- is done after reading its count.  Same arithmetic per angle as the searching kernels
// (canonical minimum-image vectors, ase get_angles order, numpy.histogram edges).
struct TrDerived {
    int32_t trip, species, off, count;
};

template <bool ORTHO>
__global__ __launch_bounds__(256) void bad_transposed_kernel(NbrFastArgs fa, const TrDerived *__restrict__ der,
                                                             const int2 *__restrict__ twork)
{
    const NbrArgs &a = fa.a;
    extern __shared__ unsigned thist[];                 // [nb] unless the counts go straight to global memory
    const int tid = threadIdx.x;
    const int2 w = twork[blockIdx.x];                   // (derived triple record, first centre of its species)
    const TrDerived dr = der[w.x];
    const int trip = dr.trip;
    const int rank = w.y + tid;
    const bool has = rank < dr.count;
    const int g = dr.off + min(rank, dr.count - 1);
    const int64_t centre = a.perm[fa.sp_first[dr.species] + min(rank, dr.count - 1)];
    const int nb = a.nb;
    const bool direct = a.cn_max > 0 || a.global_hist;
    for (int k = tid; k < nb && !direct; k += 256) thist[k] = 0u;
    __syncthreads();
    const double hb_e0 = a.edges[0], hb_en = a.edges[nb], hb_inv_w = (double)nb / (hb_en - hb_e0);
    unsigned long long nang = 0;
    const int f0 = blockIdx.y * a.frames_per_chunk, f1 = min(f0 + a.frames_per_chunk, fa.nf);
    for (int fl = f0; fl < f1; fl++) {
        const size_t slot = (size_t)fl * a.tr_total + g;
        const int n = has ? (int)a.tcount[slot] : 0;
        if (n < 2) continue;
        if (n > TR_CAP) {           // fuller than the lists: the whole call goes to the exact kernels
            a.flags[1] = 1;
            continue;
        }
        const int f = fa.f_base + fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const double *__restrict__ geo = a.geom + (size_t)(a.n_cells == 1 ? 0 : f) * GEOM_STRIDE;
        const double *pc = p + (size_t)centre * 3;
        const uint32_t *__restrict__ lst = a.tlist + slot * TR_CAP;
        const size_t hslot = a.cn_max > 0 ? (size_t)trip * (a.cn_max + 1) + min(n, a.cn_max) : (size_t)trip;
        for (int u = 0; u < n; u++) {
            const double *pu = p + (size_t)lst[u] * 3;
            double vx, vy, vz, ax, ay, az;
            pair_base<ORTHO>(geo, pu[0] - pc[0], pu[1] - pc[1], pu[2] - pc[2], vx, vy, vz);
            if (!unit_vec(vx, vy, vz, ax, ay, az)) { a.flags[0] = 1; break; }
            for (int v = u + 1; v < n; v++) {
                const double *pv = p + (size_t)lst[v] * 3;
                double wx, wy, wz, bx, by, bz;
                pair_base<ORTHO>(geo, pv[0] - pc[0], pv[1] - pc[1], pv[2] - pc[2], wx, wy, wz);
                if (!unit_vec(wx, wy, wz, bx, by, bz)) { a.flags[0] = 1; break; }
                double dot = ax * bx + ay * by + az * bz;
                if (dot > 1.0) dot = 1.0;
                if (dot < -1.0) dot = -1.0;
                const double ang = (180.0 / M_PI) * acos_fd(dot);
                const int k = hist_bin(a.edges, nb, ang, hb_e0, hb_en, hb_inv_w, a.edge_step);
                if (direct) {
                    atomicAdd(&a.n_angles[hslot], 1ull);
                    if (k >= 0) atomicAdd(&a.hist[hslot * nb + k], 1ull);
                } else {
                    nang++;
                    if (k >= 0) atomicAdd(&thist[k], 1u);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long *H = a.hist + (size_t)trip * nb;
    for (int k = tid; k < nb && !direct; k += 256) {
        const unsigned v = thist[k];
        if (v) atomicAdd(&H[k], (unsigned long long)v);
    }
    for (int off = 32; off > 0; off >>= 1) nang += __shfl_down(nang, off, 64);
    if ((tid & 63) == 0 && nang) atomicAdd(&a.n_angles[trip], nang);
}

// ---- BAD on the whole-frame tier: rows of unit vectors in HBM, every triple from the rows -------------------------------
//   lists_frame_kernel  one workgroup per (species pair, frame): sort as cn_frame_kernel, search from the species with
//                       FEWER atoms, and for every pair found compute the canonical unit vector once (the positions are
//                       still warm in L2) and append it to the centre's row and, negated, to the partner's -- the
//                       canonical arithmetic is sign-symmetric (rint, fma, sqrt, division), so the negated vector IS the
//                       one the partner-centred search would have computed.  Slots are claimed with LDS byte counters:
//                       no global atomics.  Rows: [frame][R][16] (ux, uy, uz) + [frame][R] counts; R = the centres of
//                       every ordered pair (A, B) some triple needs, indexed by rank inside species A.
//   bad_rows_kernel     one lane per centre of a triple B-A-B / X-A-X: streams its rows (one per partner species) and
//                       bins the angle of every unordered pair of unit vectors (ase get_angles: dot, clip, acos;
//                       numpy.histogram edges) into an LDS histogram.  No gathers, no barriers inside the loop.
// The 17 triples the reference asks for with three cutoffs (amof/bad.py:126-131: every ordered species pair + X) share
// three sorts and three LDS searches.
constexpr int NBRL_CAP = NBRF_NLIST;      // a fuller centre sends the call to the exact kernels, as in bad_fast_kernel
constexpr int NBRL_EW = 4;                // doubles per row entry: (ux, uy, uz, -) -- one aligned 32-byte sector per unit vector
constexpr int NBRW_HITS = 4096;           // pairs lists_frame_kernel buffers in LDS: 256 per wave, flushed by the wave itself once it
                                          // holds more than 64 -- its next 64 tasks may then add 128 (two neighbours per centre and
                                          // ROW of cells) and 64 more (third neighbours) before the call falls back to the exact kernels
constexpr int NBRW_HITS_WAVE = NBRW_HITS / (NBRW_THREADS / 64);

struct NbrListArgs {
    const int32_t *region_of;  // [S][S] first row of the ordered pair (centre species, partner species); -1: not kept
    const int32_t *inv_rank;   // [N] rank of an atom inside its species
    uint32_t *count;           // [frames of the batch][R]
    double *rows;              // [NBRL_CAP][frames of the batch * R][NBRL_EW] unit vectors centre -> neighbour, slot-major: the
                               // k-th neighbours of consecutive centres are neighbours in memory (dense writes, coalesced reads)
    size_t plane;              // frames of the batch * R
    int32_t R;
};

#ifdef NBR_PHASE_STAMPS
__device__ unsigned long long nbr_phase_ticks[16][5];   // [entry][sort, search, unit vectors, counts, workgroups] (diagnostic build only)
#define NBR_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime();
#define NBR_PHASE(k, t1, t0) if (threadIdx.x == 0) atomicAdd(&nbr_phase_ticks[blockIdx.x & 15][k], (t1) - (t0));
#else
#define NBR_STAMP(v)
#define NBR_PHASE(k, t1, t0)
#endif
template <bool ORTHO, int PT, bool COMPACT>
__global__ __launch_bounds__(NBRW_THREADS, (PT == 8 && !COMPACT) ? 4 : 8) void lists_frame_kernel(NbrArgs a, FrameArgs fr, NbrListArgs la)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ unsigned wsum[NBRW_THREADS / 64];
    FrameLds L;
    const int tid = threadIdx.x;
    const FrameItem it = fr.items[blockIdx.x];          // (sa = the species with fewer atoms: it searches)
    const int fl = (int)blockIdx.y, f = fr.f_base + fl;
    const int nA = (int)(fr.sp_first[it.sa + 1] - fr.sp_first[it.sa]);
    const int nB = it.sa == it.sb ? 0 : (int)(fr.sp_first[it.sb + 1] - fr.sp_first[it.sb]);
    constexpr bool SLAB = PT == 0;
    static_assert(!(SLAB && COMPACT), "slabs keep 16-byte records");
    L.rec = lds_raw;
    L.cell_end = reinterpret_cast<uint32_t *>(lds_raw + (size_t)(SLAB ? it.cap : nA + nB) * (COMPACT ? sizeof(uint2) : sizeof(uint4)));
    L.sidx = COMPACT ? fr.sidx + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)fr.sidx_stride : nullptr;
    const int n = SLAB ? it.cap : nA + nB, ntab = it.nx * it.ny * it.nz;
    // (a slab of a pair: a partner may sit in the halo of other slabs too, so its row's slots are claimed in global memory)
    const bool shared_partners = SLAB && it.cn != it.nzg;
    // neighbours found so far, one BYTE per atom (sorted position), four to a word: a lane claims a slot with
    // atomicAdd(word, 1 << 8 * (c & 3)) (an atom past 16 fails the call anyway, so a carry into the next byte is harmless)
    uint32_t *cnt = L.cell_end + ntab;
    // pairs found, (centre << 13 | partner) by sorted position, waiting for their unit vector: computing it inside the
    // search loop made every wave pay the float64 path on every trip (some lane always has a hit)
    uint32_t *hits = cnt + (n + 3) / 4;
    // Every wave has its own piece of the buffer and flushes it itself: no barrier between the sort and the counts.  (Round 3
    // flushed the whole buffer behind a barrier after every round of 1024 tasks: a round then lasted as long as the fullest of
    // its 1024 rows, 2.4 us -- 71 of the 94 us of a C + H frame.)
    __shared__ unsigned nh_w[NBRW_THREADS / 64];
    for (int c = tid; c < (n + 3) / 4; c += NBRW_THREADS) cnt[c] = 0u;
    if (tid < NBRW_THREADS / 64) nh_w[tid] = 0u;
    int ncen = nA, cbeg = 0, first = nB > 0 ? nA : 0, total = nA + nB;
    NBR_STAMP(ts0)
    if constexpr (SLAB) {
        __shared__ unsigned ctl[4];
        if (!frame_sort_slab(a, fr, it, L, f, nB, wsum, ctl, ncen, cbeg, first, total)) return;
    } else {
        frame_sort<PT, COMPACT>(a, fr, it, L, f, nA, nB, wsum);
    }
    NBR_STAMP(ts1)
    NBR_PHASE(0, ts1, ts0)
    const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
    const int gi = a.n_cells == 1 ? 0 : f;
    const double *__restrict__ geo = a.geom + (size_t)gi * GEOM_STRIDE;
    float sc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sc[k] = fr.cells[gi].sc[k];
    const double rc = a.cutoff[it.sa * a.S + it.sb];
    const size_t base = (size_t)fl * la.R;
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t *hw = hits + wave * NBRW_HITS_WAVE;
    unsigned *nhp = &nh_w[wave];
    auto claim = [&](int c) -> unsigned {       // next free slot of the atom at sorted position c
        const unsigned sh = 8u * ((unsigned)c & 3u);
        return (atomicAdd(&cnt[c >> 2], 1u << sh) >> sh) & 0xffu;
    };
    // one lane per buffered pair of this wave: the canonical unit vector once, to the centre's row and, negated, to the partner's
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (a wave's LDS operations execute in order: its appends are in place)
        const unsigned nh = min(__atomic_load_n(nhp, __ATOMIC_RELAXED), (unsigned)NBRW_HITS_WAVE);
        for (unsigned h = lane; h < nh; h += 64) {
            const int c = (int)(hw[h] >> 13), j = (int)(hw[h] & 0x1fffu);
            uint32_t atom_c, atom_j;
            int rank_c, rank_j;
            if (COMPACT) {
                const uint2 ic = L.sidx[c], ij = L.sidx[j];
                atom_c = ic.x; rank_c = (int)ic.y; atom_j = ij.x; rank_j = (int)ij.y;
            } else {
                atom_c = reinterpret_cast<const uint4 *>(L.rec)[c].w & ~NBRW_SLOT1;
                atom_j = reinterpret_cast<const uint4 *>(L.rec)[j].w & ~NBRW_SLOT1;
                rank_c = la.inv_rank[atom_c]; rank_j = la.inv_rank[atom_j];                // (loaded beside the positions)
            }
            const double *pc = p + (size_t)atom_c * 3, *pj = p + (size_t)atom_j * 3;
            double vx, vy, vz, ux = 0.0, uy = 0.0, uz = 0.0;
            pair_base<ORTHO>(geo, pj[0] - pc[0], pj[1] - pc[1], pj[2] - pc[2], vx, vy, vz);
            if (!unit_vec(vx, vy, vz, ux, uy, uz)) a.flags[0] = 1;          // (the call fails: results are discarded)
            if (it.reg_ab >= 0) {       // (one species: the pair is found from both ends, each end fills its own row)
                const unsigned k = claim(c);
                if (k < (unsigned)NBRL_CAP) {
                    double2 *e = reinterpret_cast<double2 *>(la.rows + ((size_t)k * la.plane + base + (size_t)(it.reg_ab + rank_c)) * NBRL_EW);
                    e[0] = make_double2(ux, uy); e[1] = make_double2(uz, 0.0);
                } else {
                    a.flags[1] = 1;
                }
            }
            if (it.reg_ba >= 0 && nB > 0) {
                const unsigned k = shared_partners ? atomicAdd(&la.count[base + (size_t)(it.reg_ba + rank_j)], 1u) : claim(j);
                if (k < (unsigned)NBRL_CAP) {
                    double2 *e = reinterpret_cast<double2 *>(la.rows + ((size_t)k * la.plane + base + (size_t)(it.reg_ba + rank_j)) * NBRL_EW);
                    e[0] = make_double2(-ux, -uy); e[1] = make_double2(-uz, 0.0);
                } else {
                    a.flags[1] = 1;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __atomic_store_n(nhp, 0u, __ATOMIC_RELAXED);
    };
    const int tasks = ncen * 9;
    for (int t0 = wave * 64; t0 < tasks; t0 += NBRW_THREADS) {
        const int t = t0 + lane;
        // a task keeps its first two pairs in registers and the wave appends them with ONE LDS atomic after the search (a
        // ballot + atomic + shuffle on every trip of the candidate loop tripled the search: 24 us against 7 in cn_frame_kernel);
        // a third pair of one centre in one row of cells is appended on the spot
        int h0 = 0, h1 = 0, mine = 0;
        const int c = cbeg + (t < tasks ? t / 9 : 0);
        if (t < tasks) {
            const int r9 = t % 9;
            frame_row_neighbours<ORTHO, COMPACT>(fr, it, L, first, sc, geo, p, c, nB == 0, frame_rec<COMPACT>(L, c), r9, rc,
                                                 [&](bool nbr, int j) {
                if (!nbr) return;
                if (mine == 0) h0 = j;
                else if (mine == 1) h1 = j;
                else {
                    const unsigned h = atomicAdd(nhp, 1u);
                    if (h < (unsigned)NBRW_HITS_WAVE) hw[h] = ((uint32_t)c << 13) | (uint32_t)j;
                    else a.flags[1] = 1;
                }
                mine++;
            });
        }
        {
            const unsigned long long m1 = __ballot(mine >= 1), m2 = __ballot(mine >= 2);
            if (m1) {
                const int leader = __ffsll((long long)m1) - 1;
                unsigned h = 0;
                if (lane == leader) h = atomicAdd(nhp, (unsigned)(__popcll(m1) + __popcll(m2)));
                const unsigned long long lt = (1ull << lane) - 1ull;
                h = __shfl(h, leader, 64) + (unsigned)(__popcll(m1 & lt) + __popcll(m2 & lt));
                if (mine >= 1) {
                    if (h < (unsigned)NBRW_HITS_WAVE) hw[h] = ((uint32_t)c << 13) | (uint32_t)h0;
                    else a.flags[1] = 1;        // (absurdly many pairs: the exact kernels take the call)
                }
                if (mine >= 2) {
                    if (h + 1 < (unsigned)NBRW_HITS_WAVE) hw[h + 1] = ((uint32_t)c << 13) | (uint32_t)h1;
                    else a.flags[1] = 1;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (__atomic_load_n(nhp, __ATOMIC_RELAXED) > 64u) flush();          // (the same word for every lane: a uniform decision)
    }
    flush();
    NBR_STAMP(tr2)
    NBR_PHASE(1, tr2, ts1)
    __syncthreads();
    NBR_STAMP(tc0)
    for (int c = tid; c < total; c += NBRW_THREADS) {
        // (one species: only the centres of the slab's core own their rows here; partners counted in global memory: nothing to store)
        const bool is_centre = nB > 0 ? c < ncen : (c >= cbeg && c < cbeg + ncen);
        if (!is_centre && (nB == 0 || shared_partners)) continue;
        const int reg = is_centre ? it.reg_ab : it.reg_ba;
        if (reg < 0) continue;
        const uint32_t k = (cnt[c >> 2] >> (8u * ((unsigned)c & 3u))) & 0xffu;
        const int rank = COMPACT ? (int)L.sidx[c].y : la.inv_rank[reinterpret_cast<const uint4 *>(L.rec)[c].w & ~NBRW_SLOT1];
        la.count[base + (size_t)(reg + rank)] = min(k, (uint32_t)NBRL_CAP);
    }
#ifdef NBR_PHASE_STAMPS
    __syncthreads();
    {
        NBR_STAMP(tc1)
        NBR_PHASE(3, tc1, tc0)
        if (threadIdx.x == 0) atomicAdd(&nbr_phase_ticks[blockIdx.x & 15][4], 1ull);
    }
#endif
}

// work item (blockIdx.y): (triple, centre species, B or -1, -); blockIdx.x strides over the tiles of 256 (frame, centre)
// pairs of the batch; one LDS histogram per workgroup, no barrier inside the loop (a lane owns its centre)
template <bool ORTHO>
__global__ __launch_bounds__(NBRF_TILE) void bad_rows_kernel(NbrArgs a, NbrListArgs la, const int4 *__restrict__ aw,
                                                             const int64_t *__restrict__ sp_first, int nf)
{
    extern __shared__ unsigned hist[];                                    // [nb] unless the counts go straight to global memory
    const int tid = threadIdx.x;
    const int4 w = aw[blockIdx.y];
    const int trip = w.x, sa = w.y, B = w.z;
    const uint32_t nA = (uint32_t)(sp_first[sa + 1] - sp_first[sa]);
    const uint32_t total = (uint32_t)nf * nA;                             // (< 2^31: the host sizes the frame batch)
    const uint32_t tiles = (total + NBRF_TILE - 1) / NBRF_TILE;
    const int nb = a.nb;
    const double hb_e0 = a.edges[0], hb_en = a.edges[nb], hb_inv_w = (double)nb / (hb_en - hb_e0);
    const bool direct = a.cn_max > 0 || a.global_hist;
    for (int k = tid; k < nb && !a.global_hist; k += NBRF_TILE) hist[k] = 0u;
    __syncthreads();
    // the partner species of the triple that keep rows for this centre species (B >= 0: one), in species order
    int n_reg = 0, reg0 = 0, reg1 = 0, reg2 = 0;
    for (int sb = 0; sb < a.S; sb++)
        if ((B < 0 || sb == B) && la.region_of[sa * a.S + sb] >= 0) {
            const int reg = la.region_of[sa * a.S + sb];
            if (n_reg == 0) reg0 = reg; else if (n_reg == 1) reg1 = reg; else if (n_reg == 2) reg2 = reg;
            n_reg++;
        }
    auto reg_sb = [&](int q) {          // species of the q-th such partner (q >= 3 only: rare)
        int seen = 0;
        for (int sb = 0; sb < a.S; sb++)
            if ((B < 0 || sb == B) && la.region_of[sa * a.S + sb] >= 0 && seen++ == q) return sb;
        return 0;
    };
    unsigned long long nang = 0;
    // a centre's rows: the counts of its first three partner species in registers, any further ones re-read
    struct Centre {
        size_t cbase;
        int n, c0, c1, c2;
    };
    auto load_centre = [&](uint32_t tile) -> Centre {
        Centre ce{0, 0, 0, 0, 0};
        const uint32_t flat = tile * (uint32_t)NBRF_TILE + tid;
        if (tile >= tiles || flat >= total) return ce;
        const uint32_t fl = flat / nA, r = flat - fl * nA;
        ce.cbase = (size_t)fl * la.R + r;
        for (int q = 0; q < n_reg; q++) {
            const int c = (int)la.count[ce.cbase + (q == 0 ? reg0 : q == 1 ? reg1 : q == 2 ? reg2 : la.region_of[sa * a.S + reg_sb(q)])];
            if (q == 0) ce.c0 = c; else if (q == 1) ce.c1 = c; else if (q == 2) ce.c2 = c;
            ce.n += c;
        }
        return ce;
    };
    auto angles_of = [&](const Centre &ce) {
        const int n = ce.n;
        if (n < 2) return;          // a centre with a single neighbour -- every N of ZIF-4 -- forms no angle
        if (n > NBRL_CAP) { a.flags[1] = 1; return; }        // (as bad_fast_kernel: more than 16 in all -> the exact kernels)
        // entry e of the centre = the e-th unit vector over its rows in species order
        auto entry = [&](int e) -> const double * {
            if (e < ce.c0) return la.rows + ((size_t)e * la.plane + ce.cbase + reg0) * NBRL_EW;
            e -= ce.c0;
            if (e < ce.c1) return la.rows + ((size_t)e * la.plane + ce.cbase + reg1) * NBRL_EW;
            e -= ce.c1;
            if (e < ce.c2) return la.rows + ((size_t)e * la.plane + ce.cbase + reg2) * NBRL_EW;
            e -= ce.c2;
            for (int q = 3; q < n_reg; q++) {
                const int reg = la.region_of[sa * a.S + reg_sb(q)];
                const int c = (int)la.count[ce.cbase + reg];
                if (e < c) return la.rows + ((size_t)e * la.plane + ce.cbase + reg) * NBRL_EW;
                e -= c;
            }
            return la.rows;     // (not reached: e < n)
        };
        const size_t hslot = a.cn_max > 0 ? (size_t)trip * (a.cn_max + 1) + min(n, a.cn_max) : (size_t)trip;
        auto bin_angle = [&](double ax, double ay, double az, double bx, double by, double bz) {
            double dot = ax * bx + ay * by + az * bz;
            if (dot > 1.0) dot = 1.0;
            if (dot < -1.0) dot = -1.0;
            const double ang = (180.0 / M_PI) * acos_fd(dot);
            const int k = hist_bin(a.edges, nb, ang, hb_e0, hb_en, hb_inv_w, a.edge_step);
            if (direct) {           // BadByCn: keyed by the number of B-neighbours of this centre
                atomicAdd(&a.n_angles[hslot], 1ull);
                if (k >= 0) atomicAdd(&a.hist[hslot * nb + k], 1ull);
            } else {
                nang++;
                if (k >= 0) atomicAdd(&hist[k], 1u);
            }
        };
        if (n_reg == 1) {
            // one partner species (every triple but X-A-X): the centre's k-th vector is one plane further
            const double2 *__restrict__ e0 = reinterpret_cast<const double2 *>(la.rows + (ce.cbase + reg0) * NBRL_EW);
            const size_t step = la.plane * (NBRL_EW / 2);
            if (n <= 4) {
                // the common case (4 N around a Zn, 2 - 3 around a C): vectors in registers, pairs unrolled
                const double2 v0a = e0[0], v0b = e0[1], v1a = e0[step], v1b = e0[step + 1];
                bin_angle(v0a.x, v0a.y, v0b.x, v1a.x, v1a.y, v1b.x);
                if (n > 2) {
                    const double2 v2a = e0[2 * step], v2b = e0[2 * step + 1];
                    bin_angle(v0a.x, v0a.y, v0b.x, v2a.x, v2a.y, v2b.x);
                    bin_angle(v1a.x, v1a.y, v1b.x, v2a.x, v2a.y, v2b.x);
                    if (n > 3) {
                        const double2 v3a = e0[3 * step], v3b = e0[3 * step + 1];
                        bin_angle(v0a.x, v0a.y, v0b.x, v3a.x, v3a.y, v3b.x);
                        bin_angle(v1a.x, v1a.y, v1b.x, v3a.x, v3a.y, v3b.x);
                        bin_angle(v2a.x, v2a.y, v2b.x, v3a.x, v3a.y, v3b.x);
                    }
                }
                return;
            }
            for (int u = 0; u + 1 < n; u++) {
                const double2 a01 = e0[u * step], a2 = e0[u * step + 1];
                for (int v = u + 1; v < n; v++) {
                    const double2 b01 = e0[v * step], b2 = e0[v * step + 1];
                    bin_angle(a01.x, a01.y, a2.x, b01.x, b01.y, b2.x);
                }
            }
            return;
        }
        for (int u = 0; u + 1 < n; u++) {
            const double2 *eu = reinterpret_cast<const double2 *>(entry(u));
            const double2 a01 = eu[0], a2 = eu[1];
            for (int v = u + 1; v < n; v++) {
                const double2 *ev = reinterpret_cast<const double2 *>(entry(v));
                const double2 b01 = ev[0], b2 = ev[1];
                bin_angle(a01.x, a01.y, a2.x, b01.x, b01.y, b2.x);
            }
        }
    };
    // two tiles per trip: the counts of the second are in flight while the first forms its angles
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += 2 * gridDim.x) {
        const Centre c_a = load_centre(tile), c_b = load_centre(tile + gridDim.x);
        angles_of(c_a);
        angles_of(c_b);
    }
    __syncthreads();
    unsigned long long *H = a.hist + (size_t)trip * nb;
    for (int k = tid; k < nb && !direct; k += NBRF_TILE) {
        const unsigned v = hist[k];
        if (v) atomicAdd(&H[k], (unsigned long long)v);
    }
    for (int off = 32; off > 0; off >>= 1) nang += __shfl_down(nang, off, 64);
    if ((tid & 63) == 0 && nang) atomicAdd(&a.n_angles[trip], nang);
}

// All triples of one centre species in ONE pass over its rows (histograms in LDS, no BadByCn keys).  The reference asks
// for every ordered species pair plus 'X' (amof/bad.py:126-131): B-A-B, X-A-X and X-X-X count the same angles again and
// again -- X-A-X is every unordered neighbour pair of an A centre, B-A-B those of them with both neighbours B, X-X-X the
// sum of X-A-X over A (X-B-X, centre any species, the sum of B-A-B over A).  Here every angle is computed once and added to
// the histogram of its partner species (if both neighbours share it) and to the all-pairs histogram; at the end the
// workgroup adds each LDS histogram to every triple it belongs to.  17 triples of three cutoffs: 14 passes and 37k angles
// a frame before, 4 passes and 14.8k angles now.
struct MergedItem {
    int32_t sa, n_reg;
    int32_t reg[3];            // first row of region (sa, b_q), q < n_reg (partner species in species order)
    int32_t tb[3][2];          // triples that take the pairs inside region q: (sa, b_q), (X, b_q); -1: not asked for
    int32_t tx[2];             // triples that take every pair: (sa, X), (X, X)
    int32_t _pad[3];
};
constexpr int NBRM_THREADS = 512;

template <bool ORTHO>
__global__ __launch_bounds__(NBRM_THREADS) void bad_rows_merged_kernel(NbrArgs a, NbrListArgs la, const MergedItem *__restrict__ items,
                                                                       const int64_t *__restrict__ sp_first, int nf)
{
    extern __shared__ unsigned hist[];          // [n_reg (+ 1 when there are several partner species)][nb]
    const int tid = threadIdx.x;
    const MergedItem it = items[blockIdx.y];
    const uint32_t nA = (uint32_t)(sp_first[it.sa + 1] - sp_first[it.sa]);
    const uint32_t total = (uint32_t)nf * nA;                             // (< 2^31: the host sizes the frame batch)
    const uint32_t tiles = (total + NBRM_THREADS - 1) / NBRM_THREADS;
    const int nb = a.nb, n_reg = it.n_reg;
    const bool all_pairs = n_reg > 1 && (it.tx[0] >= 0 || it.tx[1] >= 0);       // (one partner species: its histogram IS the all-pairs one)
    const int n_hist = n_reg + (n_reg > 1 ? 1 : 0);
    unsigned *hist_x = hist + (size_t)n_reg * nb;
    const double hb_e0 = a.edges[0], hb_en = a.edges[nb], hb_inv_w = (double)nb / (hb_en - hb_e0);
    for (int k = tid; k < n_hist * nb; k += NBRM_THREADS) hist[k] = 0u;
    __syncthreads();
    unsigned long long nang0 = 0, nang1 = 0, nang2 = 0, nangx = 0;
    const int reg0 = it.reg[0], reg1 = it.reg[1], reg2 = it.reg[2];
    const bool w0 = it.tb[0][0] >= 0 || it.tb[0][1] >= 0 || n_reg == 1, w1 = n_reg > 1 && (it.tb[1][0] >= 0 || it.tb[1][1] >= 0),
               w2 = n_reg > 2 && (it.tb[2][0] >= 0 || it.tb[2][1] >= 0);
    struct Centre {
        size_t cbase;
        int c0, c1, c2;
    };
    auto load_centre = [&](uint32_t tile) -> Centre {
        Centre ce{0, 0, 0, 0};
        const uint32_t flat = tile * (uint32_t)NBRM_THREADS + tid;
        if (tile >= tiles || flat >= total) return ce;
        const uint32_t fl = flat / nA, r = flat - fl * nA;
        ce.cbase = (size_t)fl * la.R + r;
        ce.c0 = (int)la.count[ce.cbase + reg0];
        if (n_reg > 1) ce.c1 = (int)la.count[ce.cbase + reg1];
        if (n_reg > 2) ce.c2 = (int)la.count[ce.cbase + reg2];
        return ce;
    };
    auto bin_of = [&](const double2 &a01, const double2 &a2, const double2 &b01, const double2 &b2) -> int {
        double dot = a01.x * b01.x + a01.y * b01.y + a2.x * b2.x;
        if (dot > 1.0) dot = 1.0;
        if (dot < -1.0) dot = -1.0;
        return hist_bin(a.edges, nb, (180.0 / M_PI) * acos_fd(dot), hb_e0, hb_en, hb_inv_w, a.edge_step);
    };
    auto angles_of = [&](const Centre &ce) {
        const int n = ce.c0 + ce.c1 + ce.c2;
        if (n < 2) return;
        const int o1 = ce.c0, o2 = ce.c0 + ce.c1;
        auto entry = [&](int e) -> const double2 * {     // the e-th unit vector of the centre over its rows in species order
            const int q = e < o1 ? 0 : (e < o2 ? 1 : 2);
            const int i = e - (q == 0 ? 0 : (q == 1 ? o1 : o2)), reg = q == 0 ? reg0 : (q == 1 ? reg1 : reg2);
            return reinterpret_cast<const double2 *>(la.rows + ((size_t)i * la.plane + ce.cbase + reg) * NBRL_EW);
        };
        if (all_pairs) {
            if (n > NBRL_CAP) { a.flags[1] = 1; return; }       // (as bad_fast_kernel: more than 16 in all -> the exact kernels)
            for (int u = 0; u + 1 < n; u++) {
                const double2 *eu = entry(u);
                const double2 a01 = eu[0], a2 = eu[1];
                const int qu = u < o1 ? 0 : (u < o2 ? 1 : 2);
                for (int v = u + 1; v < n; v++) {
                    const double2 *ev = entry(v);
                    const int k = bin_of(a01, a2, ev[0], ev[1]);
                    const int qv = v < o1 ? 0 : (v < o2 ? 1 : 2);
                    nangx++;
                    if (k >= 0) atomicAdd(&hist_x[k], 1u);
                    if (qu == qv && (qu == 0 ? w0 : (qu == 1 ? w1 : w2))) {
                        if (qu == 0) nang0++; else if (qu == 1) nang1++; else nang2++;
                        if (k >= 0) atomicAdd(&hist[(size_t)qu * nb + k], 1u);
                    }
                }
            }
            return;
        }
        // only the pairs inside a partner species are asked for
        for (int q = 0; q < n_reg; q++) {
            const int c = q == 0 ? ce.c0 : (q == 1 ? ce.c1 : ce.c2);
            if (c < 2 || !(q == 0 ? w0 : (q == 1 ? w1 : w2))) continue;
            const int off = q == 0 ? 0 : (q == 1 ? o1 : o2);
            for (int u = 0; u + 1 < c; u++) {
                const double2 *eu = entry(off + u);
                const double2 a01 = eu[0], a2 = eu[1];
                for (int v = u + 1; v < c; v++) {
                    const double2 *ev = entry(off + v);
                    const int k = bin_of(a01, a2, ev[0], ev[1]);
                    if (q == 0) nang0++; else if (q == 1) nang1++; else nang2++;
                    if (k >= 0) atomicAdd(&hist[(size_t)q * nb + k], 1u);
                }
            }
        }
    };
    // (measured and rejected, round 4: a centre's unit vectors fetched into registers in one round trip and the pairs formed
    //  from registers, fully unrolled -- 2.6 -> 3.0 ms at four vectors, 3.7 ms at eight: the loop is not waiting for them)
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += 2 * gridDim.x) {
        const Centre c_a = load_centre(tile), c_b = load_centre(tile + gridDim.x);
        angles_of(c_a);
        angles_of(c_b);
    }
    __syncthreads();
    // every LDS histogram to every triple it belongs to
    auto flush = [&](const unsigned *h, unsigned long long nang, int t0, int t1) {
        for (int off = 32; off > 0; off >>= 1) nang += __shfl_down(nang, off, 64);
        for (int w = 0; w < 2; w++) {
            const int trip = w == 0 ? t0 : t1;
            if (trip < 0) continue;
            unsigned long long *H = a.hist + (size_t)trip * nb;
            for (int k = tid; k < nb; k += NBRM_THREADS) {
                const unsigned v = h[k];
                if (v) atomicAdd(&H[k], (unsigned long long)v);
            }
            if ((tid & 63) == 0 && nang) atomicAdd(&a.n_angles[trip], nang);
        }
    };
    flush(hist, nang0, it.tb[0][0], it.tb[0][1]);
    if (n_reg > 1) flush(hist + nb, nang1, it.tb[1][0], it.tb[1][1]);
    if (n_reg > 2) flush(hist + 2 * (size_t)nb, nang2, it.tb[2][0], it.tb[2][1]);
    if (n_reg == 1) flush(hist, nang0, it.tx[0], it.tx[1]);
    else flush(hist_x, nangx, it.tx[0], it.tx[1]);
}

// ------------------------------------------------------------ host side ----
struct NbrSetup {
    HostGeom geom;
    std::vector<double> img;
    std::vector<int32_t> nimg;
    int max_img = 0;
    HostTiles tiles;
    NbrArgs a;
    Stager stage;
};

static int nbr_setup(amof_ctx *ctx, const amof_traj *t, const double *cutoff, int tile, NbrSetup &s)
{
    const int S = t->n_species;
    double R = 0.0;
    for (int k = 0; k < S * S; k++) {
        if (!(cutoff[k] >= 0.0) || !isfinite(cutoff[k])) return fail(ctx, AMOF_EINVAL, "cutoff must be finite and >= 0");
        R = std::max(R, cutoff[k]);
    }
    for (int x = 0; x < S; x++)
        for (int y = 0; y < S; y++)
            if (cutoff[x * S + y] != cutoff[y * S + x]) return fail(ctx, AMOF_EINVAL, "cutoff matrix must be symmetric");
    AMOF_TRY(build_geometry(ctx, t, s.geom));
    AMOF_TRY(build_images(ctx, t, s.geom, R, s.img, s.nimg, s.max_img));
    build_tiles(t, tile, s.tiles);
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    AMOF_TRY(stager_begin(ctx, t, true, s.stage));
    const double *pos_dev = s.stage.dev;
    UploadPack pk;      // (one copy for the eight tables)
    const int i_geom = pk.add(s.geom.rec.data(), s.geom.rec.size() * sizeof(double));
    const int i_img = pk.add(s.img.data(), s.img.size() * sizeof(double));
    const int i_nimg = pk.add(s.nimg.data(), s.nimg.size() * sizeof(int32_t));
    const int i_perm = pk.add(s.tiles.perm.data(), s.tiles.perm.size() * sizeof(int32_t));
    const int i_tiles = pk.add(s.tiles.tiles.data(), s.tiles.tiles.size() * sizeof(Tile));
    const int i_ft = pk.add(s.tiles.sp_first_tile.data(), S * sizeof(int32_t));
    const int i_nt = pk.add(s.tiles.sp_ntiles.data(), S * sizeof(int32_t));
    const int i_cut = pk.add(cutoff, (size_t)S * S * sizeof(double));
    AMOF_TRY(upload_pack(ctx, SLOT_GEOM, pk));
    NbrArgs &a = s.a;
    a = NbrArgs{};
    a.pos = pos_dev;
    a.geom = pk.ptr<double>(i_geom);
    a.img = pk.ptr<double>(i_img);
    a.nimg = pk.ptr<int32_t>(i_nimg);
    a.perm = pk.ptr<int32_t>(i_perm);
    a.tiles = pk.ptr<Tile>(i_tiles);
    a.sp_first_tile = pk.ptr<int32_t>(i_ft);
    a.sp_ntiles = pk.ptr<int32_t>(i_nt);
    a.cutoff = pk.ptr<double>(i_cut);
    a.N = t->n_atoms;
    a.F = (int32_t)t->n_frames;
    a.n_cells = (int32_t)t->n_cells;
    a.S = S;
    a.max_img = s.max_img;
    return AMOF_OK;
}

static void pick_chunks(int64_t F, size_t nwork, int32_t &fpc, unsigned &chunks)
{
    int64_t want = (4 * 2048 + (int64_t)nwork - 1) / (int64_t)std::max<size_t>(1, nwork);
    int64_t f = std::max<int64_t>(1, F / std::max<int64_t>(1, want));
    f = std::min<int64_t>(f, 64);
    int64_t c = (F + f - 1) / f;
    if (c > 65535) {
        f = (F + 65534) / 65535;
        c = (F + f - 1) / f;
    }
    fpc = (int32_t)f;
    chunks = (unsigned)c;
}

__global__ void add_u64_kernel(unsigned long long *dst, const unsigned long long *src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

// ---- fast-path preparation shared by CN and BAD ----
struct NbrFast {
    bool ok = false;
    int axis = 0;
    bool ortho = false;
    int64_t FB = 0, FB0 = 0;     // frames per batch (largest, first)
    void *d_Q = nullptr, *d_slab = nullptr, *d_cells = nullptr, *d_spfirst = nullptr, *d_qflag = nullptr;
    std::vector<int64_t> sp_first;
    NbrFastArgs fa;
    // 3-D cell list (cutoffs far below the cell size): frames sorted by (species, cell) instead of slabs
    bool cell = false;
    int nk[3] = {0, 0, 0};
    void *d_start3 = nullptr;
    unsigned long long used_mask = ~0ull;   // species with a cutoff to any species: the only ones the cell sort handles
    int64_t max_used_atoms = 0;             // (its LDS record cache is sized for the largest of them)
};

// quantise + sort one frame batch for the neighbour kernels (slab list or cell list) and point fa at it
static int nbr_fast_batch(amof_ctx *ctx, const amof_traj *t, NbrSetup &st, NbrFast &nf, int64_t fb, int64_t nfr)
{
    const NbrArgs &a = st.a;
    if (nf.cell)
        AMOF_TRY(launch_quantize_cells(ctx, a.pos, a.geom, (int)t->n_cells, a.perm, (const int64_t *)nf.d_spfirst,
                                       t->n_species, t->n_atoms, (int)fb, (int)nfr, nf.nk[0], nf.nk[1], nf.nk[2],
                                       (QAtom *)nf.d_Q, (uint32_t *)nf.d_start3, (int32_t *)nf.d_qflag, nf.max_used_atoms,
                                       nf.used_mask));
    else
        AMOF_TRY(launch_quantize(ctx, a.pos, a.geom, (int)t->n_cells, a.perm, (const int64_t *)nf.d_spfirst, t->n_species,
                                 t->n_atoms, (int)fb, (int)nfr, nf.axis, (QAtom *)nf.d_Q, (uint32_t *)nf.d_slab,
                                 (int32_t *)nf.d_qflag));
    nf.fa.f_base = (int32_t)fb;
    nf.fa.nf = (int32_t)nfr;
    return AMOF_OK;
}

static int nbr_fast_prepare(amof_ctx *ctx, const amof_traj *t, const double *cutoff, NbrSetup &st, NbrFast &nf,
                            unsigned long long centre_mask = 0ull)
{
    const int S = t->n_species;
    const int64_t nc = t->n_cells;
    const char *force = getenv("AMOF_NBR_KERNEL");
    double R = 0.0;
    for (int k = 0; k < S * S; k++) R = std::max(R, cutoff[k]);
    nf.ok = st.max_img == 0 && t->pbc[0] && t->pbc[1] && t->pbc[2] && R > 0.0 && t->n_atoms > 0 &&
            !(force && strcmp(force, "v1") == 0);
    if (!nf.ok) return AMOF_OK;
    nf.ortho = st.geom.all_ortho;
    double hmin[3] = {1e300, 1e300, 1e300}, csum = 0.0;
    for (int64_t k = 0; k < nc; k++) {
        const double *c = t->cell + 9 * k;
        for (int x = 0; x < 3; x++) hmin[x] = std::min(hmin[x], st.geom.rec[(size_t)k * GEOM_STRIDE + 18 + x]);
        csum = std::max(csum, sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) +
                                  sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5]) +
                                  sqrt(c[6] * c[6] + c[7] * c[7] + c[8] * c[8]));
    }
    nf.axis = 0;
    for (int x = 1; x < 3; x++)
        if (hmin[x] > hmin[nf.axis]) nf.axis = x;
    const int ord[3] = {(nf.axis + 1) % 3, (nf.axis + 2) % 3, nf.axis};
    const double two32 = 1.0 / 4294967296.0;
    std::vector<NbrCell> cells((size_t)nc);
    for (int64_t k = 0; k < nc; k++) {
        const double *c = t->cell + 9 * k;
        NbrCell &r = cells[(size_t)k];
        for (int q = 0; q < 9; q++) r.sc[q] = 0.f;
        if (nf.ortho) {
            for (int q = 0; q < 3; q++) r.sc[q] = (float)(c[4 * ord[q]] * two32);
        } else {
            for (int q = 0; q < 3; q++)
                for (int x = 0; x < 3; x++) r.sc[3 * q + x] = (float)(c[3 * ord[q] + x] * two32);
        }
        r._pad = 0.f;
        r.gap_per_len = 4294967296.0 / st.geom.rec[(size_t)k * GEOM_STRIDE + 18 + nf.axis] * (1.0 + 1e-6);
    }
    nf.sp_first.assign(S + 1, 0);
    for (int x = 0; x < S; x++) nf.sp_first[x + 1] = nf.sp_first[x] + st.tiles.nsp[x];
    // species without a cutoff to any species are neither centres nor partners: the cell sort skips them
    nf.used_mask = 0ull;
    nf.max_used_atoms = 1;
    for (int x = 0; x < S && x < 64; x++) {
        bool used = false;
        for (int y = 0; y < S; y++) used |= cutoff[x * S + y] > 0.0 || cutoff[y * S + x] > 0.0;
        // (centre_mask: species the caller queues as centres whatever their cutoffs -- per-atom counts of a
        //  zero-cutoff set still read the centre's record, which must therefore be sorted)
        if (used || (centre_mask >> x & 1ull)) {
            nf.used_mask |= 1ull << x;
            nf.max_used_atoms = std::max<int64_t>(nf.max_used_atoms, st.tiles.nsp[x]);
        }
    }
    if (S > 64) { nf.used_mask = ~0ull; nf.max_used_atoms = *std::max_element(st.tiles.nsp.begin(), st.tiles.nsp.end()); }
    // 3-D cell list instead of the slab list when the cutoffs are far below the cell size: cells at least R thick
    // (R = the largest cutoff), >= 3 per axis, reach 1 -- a centre meets the partners of 27 cells instead of a slab
    // range (ZIF-4 3x3x4, Zn-N at 2.5 A: ~7 candidates per Zn instead of ~930)
    {
        bool cell_ok = t->n_atoms >= 256 && t->n_atoms < (1ll << CELL_SPECIES_SHIFT) && S <= 64 && !getenv("AMOF_NBR_NOCELL");
        for (int x = 0; x < 3; x++) {
            nf.nk[x] = (int)std::min(1024.0, floor(hmin[x] / (R * (1.0 + 1e-5))));
            if (nf.nk[x] < 3) cell_ok = false;
        }
        if (cell_ok) {
            // no point in cells emptier than ~2 atoms (thicker cells stay correct, the tables shrink)
            while ((int64_t)nf.nk[0] * nf.nk[1] * nf.nk[2] > std::min<int64_t>(CELL_LDS_MAX, std::max<int64_t>(27, t->n_atoms / 2))) {
                int big = 0;
                for (int x = 1; x < 3; x++)
                    if (nf.nk[x] > nf.nk[big]) big = x;
                if (nf.nk[big] <= 3) { cell_ok = false; break; }     // (a tiny box with absurdly many atoms)
                nf.nk[big]--;
            }
            // visited share of a partner species: 27 cells of the grid vs the slab range of a 256-centre tile
            const double f3 = 27.0 / ((double)nf.nk[0] * nf.nk[1] * nf.nk[2]);
            const double f1 = std::min(1.0, 2.0 * R / hmin[nf.axis] + 0.1);
            if (!(f3 < 0.25 * f1) && !getenv("AMOF_NBR_FORCE_CELL")) cell_ok = false;
            if ((int64_t)nf.nk[0] * nf.nk[1] * nf.nk[2] * S > 0x3fffffff) cell_ok = false;
        }
        nf.cell = cell_ok;
    }
    size_t per_frame = (size_t)t->n_atoms * sizeof(QAtom);
    const int64_t nkeys = nf.cell ? (int64_t)nf.nk[0] * nf.nk[1] * nf.nk[2] * S : 0;
    if (nf.cell) per_frame += (size_t)(nkeys + 1) * sizeof(uint32_t);
    int64_t FB = std::max<int64_t>(1, (int64_t)(2ll << 30) / (int64_t)std::max<size_t>(1, per_frame));   // <= 2 GiB of scratch
    nf.FB = std::min<int64_t>(std::min<int64_t>(FB, 32768), std::max<int64_t>(1, t->n_frames));
    // host-resident input: batches of 512, 1024, 2048 ... frames, the copy of the next one overlaps this one's kernels
    nf.FB0 = st.stage.lazy ? std::min<int64_t>(nf.FB, 512) : nf.FB;
    if (nf.cell) {
        // the cell kernels read the fixed-point components in the cell's own axis order
        for (int64_t k = 0; k < nc; k++) {
            const double *c = t->cell + 9 * k;
            NbrCell &r = cells[(size_t)k];
            for (int q = 0; q < 9; q++) r.sc[q] = 0.f;
            if (nf.ortho) {
                for (int q = 0; q < 3; q++) r.sc[q] = (float)(c[4 * q] * two32);
            } else {
                for (int q = 0; q < 9; q++) r.sc[q] = (float)(c[q] * two32);
            }
        }
    }
    AMOF_TRY(upload(ctx, SLOT_AUX4, cells.data(), cells.size() * sizeof(NbrCell), &nf.d_cells));
    AMOF_TRY(upload(ctx, SLOT_AUX5, nf.sp_first.data(), nf.sp_first.size() * sizeof(int64_t), &nf.d_spfirst));
    AMOF_TRY(ensure(ctx, SLOT_HISTU, (size_t)nf.FB * t->n_atoms * sizeof(QAtom), &nf.d_Q));
    if (nf.cell) {
        AMOF_TRY(ensure(ctx, SLOT_SELF, (size_t)nf.FB * (nkeys + 1) * sizeof(uint32_t), &nf.d_start3));
    } else {
        AMOF_TRY(ensure(ctx, SLOT_SELF, (size_t)nf.FB * S * (QSLABS + 1) * sizeof(uint32_t), &nf.d_slab));
    }
    AMOF_TRY(ensure(ctx, SLOT_SPEC, sizeof(int32_t), &nf.d_qflag));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(nf.d_qflag, 0, sizeof(int32_t), ctx->stream));
    NbrFastArgs &fa = nf.fa;
    fa.a = st.a;
    fa.Q = (const QAtom *)nf.d_Q;
    fa.slab_start = (const uint32_t *)nf.d_slab;
    fa.start3 = (const uint32_t *)nf.d_start3;
    fa.nx = nf.nk[0]; fa.ny = nf.nk[1]; fa.nz = nf.nk[2];
    fa.cells = (const NbrCell *)nf.d_cells;
    fa.sp_first = (const int64_t *)nf.d_spfirst;
    // f32 chain error: fast_guard_rel (amof_internal.h); the fixed-point grid moves a distance by
    // < csum * 2^-32 (x2 margin)
    const double grel = fast_guard_rel(st.geom, nc);
    // + 3u: the cutoff rounded to f32 (1u) and the roundings of r_in / r_out = rc -+ g in the kernels
    {
        const double want = grel + 3.0 / 16777216.0;
        float fg = (float)want;
        if ((double)fg < want) fg = nextafterf(fg, INFINITY);
        fa.guard_rel = fg;
    }
    if (!nf.ortho) {
        // sheared cells: wrapped and canonical images may differ at |s_k| = 1/2 -- both must then be
        // decisively beyond every cutoff (see rdf.hip)
        for (int x = 0; x < 3; x++)
            if (R * (1.0 + 4.0 * grel + 1e-6) >= 0.5 * hmin[x]) nf.ok = false;
        if (!nf.ok) return AMOF_OK;
    }
    fa.guard_abs = (float)(csum * (1.0 / 2147483648.0));
    return AMOF_OK;
}

// ---- whole-frame-in-LDS tier: per-item grids and the few constants the kernels need ----
struct NbrFrame {
    bool ok = false;
    bool ortho = false;
    std::vector<FrameItem> items;
    size_t lds = 0;
    bool compact = false;   // 8-byte records + (atom, rank) in global scratch: pairs that 16-byte records leave one workgroup per CU
    FrameArgs fr;
    std::vector<NbrCell> cells;
    std::vector<int64_t> sp_first;
    UploadPack pk;          // cells | sp_first | items (| whatever the caller adds first): one copy (nbr_frame_commit)
    const int64_t *d_spfirst = nullptr;
    void *d_qflag = nullptr;
};

// a grid for one species pair: cells at least rc thick, >= 3 per axis, as fine as the LDS left beside n records allows
static bool frame_item_grid(const double hmin[3], double rc, int64_t n, int slots, int64_t densest, FrameItem &it, size_t &lds,
                            size_t rec_size, size_t extra_bytes = 0)
{
    if (n > NBRW_MAX_ATOMS || n <= 0) return false;
    int nk[3];
    // A centre only looks at the 27 cells around its own, so a cell must be at least rc thick IN THE COORDINATES THE CELL KEYS
    // ARE TAKEN FROM.  16-byte records key the 32-bit fixed point (2^-32 of a cell vector: inside the 1e-5 margin); COMPACT
    // 8-byte records key the coordinate TRUNCATED to 16 bits, which moves a centre against its partner by up to 2^-16 of the
    // cell vector -- far more than 1e-5 * rc / h -- so their cells are 2^-15 (twice that, in fractions of the axis) thicker.
    const double trunc = rec_size < sizeof(uint4) ? 1.0 / 32768.0 : 0.0;
    for (int x = 0; x < 3; x++) {
        nk[x] = (int)std::min(1024.0, floor(1.0 / (rc * (1.0 + 1e-5) / hmin[x] + trunc)));
        if (nk[x] < 3) return false;
    }
    const size_t rec_bytes = (size_t)n * rec_size + extra_bytes;           // (+ whatever else the kernel keeps per atom)
    // two workgroups per CU when the records leave room for a useful table, one otherwise
    size_t budget = 76 * 1024;
    if (rec_bytes + (size_t)slots * 4 * std::min<int64_t>(densest / 4 + 27, 1024) > budget) budget = 152 * 1024;
    if (rec_bytes + (size_t)slots * 4 * 27 > budget) return false;
    const int64_t cells_budget = (int64_t)((budget - rec_bytes) / ((size_t)slots * 4));
    // as fine as the budget allows (the cutoff bounds it from the other side): a centre walks 27 cells whatever their size,
    // so emptier cells are fewer candidates, and clearing + scanning 8k counters costs a thousand lanes eight steps
    const int64_t want = std::max<int64_t>(27, std::min<int64_t>(cells_budget, std::max<int64_t>(4 * densest, 4096)));
    while ((int64_t)nk[0] * nk[1] * nk[2] > want) {
        int big = 0;
        for (int x = 1; x < 3; x++)
            if (nk[x] > nk[big]) big = x;
        if (nk[big] <= 3) break;
        nk[big]--;
    }
    if ((int64_t)nk[0] * nk[1] * nk[2] > cells_budget) return false;
    it.nx = nk[0]; it.ny = nk[1]; it.nz = nk[2];
    it.zoff = 0; it.nzg = nk[2]; it.c0 = 0; it.cn = nk[2]; it.cap = (int32_t)n;
    lds = std::max(lds, rec_bytes + (size_t)slots * 4 * (size_t)nk[0] * nk[1] * nk[2]);
    return true;
}

// z-slab entries of one species pair for the streaming kernels (PT = 0, 16-byte records): the fewest slabs whose records
// (the expected share of the atoms + 20 % + 128: a denser slab raises the overflow flag and the gather kernels answer) and
// a useful table fit `budget` bytes of LDS beside `fixed` bytes and `per_atom` bytes per record.  nB = 0: one species.
// thin = false asks for at least two cells per expected partner (or the finest grid the cutoff allows).
static bool frame_item_slabs(const double hmin[3], double rc, int64_t nA, int64_t nB, const FrameItem &proto, std::vector<FrameItem> &out,
                             size_t &lds, size_t budget, size_t fixed, size_t per_atom, bool thin)
{
    int nk0[3];
    for (int x = 0; x < 3; x++) {
        nk0[x] = (int)std::min(1024.0, floor(1.0 / (rc * (1.0 + 1e-5) / hmin[x])));
        if (nk0[x] < 3) return false;
    }
    const int64_t partners = nB > 0 ? nB : nA;
    {   // the whole frame's granularity first, as frame_item_grid: about four cells per partner
        const int64_t want = std::max<int64_t>(27, std::max<int64_t>(4 * partners, 4096));
        while ((int64_t)nk0[0] * nk0[1] * nk0[2] > want) {
            int big = 0;
            for (int x = 1; x < 3; x++)
                if (nk0[x] > nk0[big]) big = x;
            if (nk0[big] <= 3) break;
            nk0[big]--;
        }
    }
    const size_t rec = sizeof(uint4) + per_atom;
    for (int nsl = 1; nsl == 1 || (nk0[2] >= 4 && nsl <= nk0[2] / 2); nsl++) {
        int nk[3] = {nk0[0], nk0[1], nk0[2]};
        const int zn = (nk[2] + nsl - 1) / nsl;
        if (nsl > 1 && zn + 2 > nk[2]) continue;
        if (nsl > 1 && (nsl - 1) * zn >= nk[2]) continue;           // (the last slab would be empty: a smaller count covers it)
        const int nzl = nsl == 1 ? nk[2] : zn + 2;
        const double share_c = nsl == 1 ? 1.0 : (double)zn / nk[2], share_p = nsl == 1 ? 1.0 : (double)nzl / nk[2];
        const int64_t est_p = (int64_t)ceil((double)partners * share_p);
        const int64_t est = nB > 0 ? (int64_t)ceil((double)nA * share_c) + est_p : est_p;
        const int64_t need = nsl == 1 ? est : est + est / 5 + 128;
        if (need > NBRW_MAX_ATOMS) continue;
        if (fixed + (size_t)need * rec + 4 * 27 > budget) continue;
        const int64_t cells_budget = (int64_t)((budget - fixed - (size_t)need * rec) / 4);
        while ((int64_t)nk[0] * nk[1] * nzl > cells_budget) {       // coarser in x / y (the slabs keep their layers)
            const int big = nk[1] > nk[0] ? 1 : 0;
            if (nk[big] <= 3) break;
            nk[big]--;
        }
        const int64_t cells = (int64_t)nk[0] * nk[1] * nzl;
        if (cells > cells_budget) continue;
        const bool finest = nk[0] == nk0[0] && nk[1] == nk0[1];
        if (!thin && !finest && cells < 2 * est_p) continue;
        const int64_t cap = std::min<int64_t>(NBRW_MAX_ATOMS, (int64_t)((budget - fixed - 4 * (size_t)cells) / rec));
        for (int sl = 0; sl < nsl; sl++) {
            FrameItem it = proto;
            it.nx = nk[0]; it.ny = nk[1]; it.nzg = nk[2];
            if (nsl == 1) {
                it.nz = nk[2]; it.zoff = 0; it.c0 = 0; it.cn = nk[2];
            } else {
                const int z0 = sl * zn, cn = std::min(zn, nk[2] - z0);
                it.nz = cn + 2; it.zoff = (z0 - 1 + nk[2]) % nk[2]; it.c0 = 1; it.cn = cn;
            }
            it.cap = (int32_t)cap;
            out.push_back(it);
        }
        lds = std::max(lds, fixed + (size_t)cap * rec + 4 * (size_t)cells);
        return true;
    }
    return false;
}

// the whole call in slabs: budget of two workgroups per CU with useful tables, then one per CU, then whatever fits.
// items_of(budget, thin, out, lds) builds every pair's entries and says whether all fit.
template <typename Build>
static bool frame_slab_passes(Build &&build, std::vector<FrameItem> &items, size_t &lds)
{
    const size_t budgets[3] = {76 * 1024, 152 * 1024, 152 * 1024};
    for (int pass = 0; pass < 3; pass++) {
        items.clear();
        lds = 0;
        if (build(budgets[pass], pass == 2, items, lds)) return true;
    }
    return false;
}
static bool frame_slabs_forced() { const char *e = getenv("AMOF_NBR_SLABS"); return e && e[0] == '1'; }
static bool frame_slabs_forbidden() { const char *e = getenv("AMOF_NBR_SLABS"); return e && e[0] == '0'; }

// the slab kernels' input: room for FB frames of records (at most 2 GiB: FB shrinks) and their bin tables
static int frame_slab_buffers(amof_ctx *ctx, const amof_traj *t, NbrFrame &nw, int64_t &FB)
{
    FB = std::max<int64_t>(1, std::min<int64_t>(FB, (int64_t)(((size_t)2 << 30) / ((size_t)t->n_atoms * sizeof(QAtom)))));
    void *d_Q, *d_z;
    AMOF_TRY(ensure(ctx, SLOT_HISTU, (size_t)FB * (size_t)t->n_atoms * sizeof(QAtom), &d_Q));
    AMOF_TRY(ensure(ctx, SLOT_SELF, (size_t)FB * (size_t)t->n_species * (QSLABS + 1) * sizeof(uint32_t), &d_z));
    nw.fr.Q = (const QAtom *)d_Q;
    nw.fr.zstart = (const uint32_t *)d_z;
    return AMOF_OK;
}
static int frame_slab_quantize(amof_ctx *ctx, const amof_traj *t, const NbrArgs &a, NbrFrame &nw, int64_t fb, int64_t nfr)
{
    unsigned long long used = 0ull;         // only the species some pair of the call reads
    for (const FrameItem &it : nw.items) {
        used |= it.sa < 64 ? 1ull << it.sa : 0ull;
        used |= it.sb < 64 ? 1ull << it.sb : 0ull;
    }
    return launch_quantize(ctx, a.pos, a.geom, (int)t->n_cells, a.perm, nw.d_spfirst, t->n_species, t->n_atoms, (int)fb, (int)nfr, 2,
                           (QAtom *)nw.fr.Q, (uint32_t *)nw.fr.zstart, (int32_t *)nw.d_qflag, 0, 1, nullptr, used);
}

// Record size of the tier, three passes: 0 = 16-byte records, 1 = COMPACT 8-byte ones, 2 = 16-byte again.  AMOF_NBR_COMPACT=1
// starts at (and keeps) the compact pass -- tests of the 16-bit cell keys; =0 never leaves pass 0.
static int frame_first_pass()
{
    const char *e = getenv("AMOF_NBR_COMPACT");
    return e && e[0] == '1' ? 1 : 0;
}
static bool frame_pass_done(int pass, bool ok, size_t lds, bool ok16)
{
    const char *e = getenv("AMOF_NBR_COMPACT");
    if (e && (e[0] == '1' || e[0] == '0')) return true;     // forced: whatever this pass gave
    if (ok && lds <= 76 * 1024) return true;                // two workgroups per CU: nothing to gain
    if (pass == 1 && ok && !ok16) return true;              // only the compact records fit at all: keep them (pass 2 would fail again)
    return pass == 2;
}

// host-side constants of the tier (nothing is uploaded before nbr_frame_commit); ok stays false when the tier cannot take the call
static int nbr_frame_prepare(const amof_traj *t, const double *cutoff, NbrSetup &st, NbrFrame &nw, double hmin[3])
{
    const int S = t->n_species;
    const int64_t nc = t->n_cells;
    double R = 0.0, csum = 0.0;
    for (int k = 0; k < S * S; k++) R = std::max(R, cutoff[k]);
    nw.ok = st.max_img == 0 && t->pbc[0] && t->pbc[1] && t->pbc[2] && R > 0.0 && t->n_atoms > 0 &&
            t->n_atoms < (1ll << CELL_SPECIES_SHIFT) && !getenv("AMOF_NBR_NOFRAME") &&
            !getenv("AMOF_NBR_NOCELL") && !getenv("AMOF_NBR_FORCE_CELL") &&      // (those name the gather kernels)
            !(getenv("AMOF_NBR_KERNEL") && strcmp(getenv("AMOF_NBR_KERNEL"), "v1") == 0);
    for (int x = 0; x < 3; x++) hmin[x] = 1e300;
    for (int64_t k = 0; k < nc; k++) {
        const double *c = t->cell + 9 * k;
        for (int x = 0; x < 3; x++) hmin[x] = std::min(hmin[x], st.geom.rec[(size_t)k * GEOM_STRIDE + 18 + x]);
        csum = std::max(csum, sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5]) +
                                  sqrt(c[6] * c[6] + c[7] * c[7] + c[8] * c[8]));
    }
    if (!nw.ok) return AMOF_OK;
    nw.ortho = st.geom.all_ortho;
    const double grel = fast_guard_rel(st.geom, nc);
    if (!nw.ortho)
        for (int x = 0; x < 3; x++)
            if (R * (1.0 + 4.0 * grel + 1e-6) >= 0.5 * hmin[x]) nw.ok = false;     // (as in nbr_fast_prepare)
    if (!nw.ok) return AMOF_OK;
    const double two32 = 1.0 / 4294967296.0;
    nw.cells.assign((size_t)nc, NbrCell{});
    for (int64_t k = 0; k < nc; k++) {
        const double *c = t->cell + 9 * k;
        NbrCell &r = nw.cells[(size_t)k];
        for (int q = 0; q < 9; q++) r.sc[q] = 0.f;
        if (nw.ortho) {
            for (int q = 0; q < 3; q++) r.sc[q] = (float)(c[4 * q] * two32);
        } else {
            for (int q = 0; q < 9; q++) r.sc[q] = (float)(c[q] * two32);
        }
        r._pad = 0.f;
        r.gap_per_len = 0.0;
    }
    nw.sp_first.assign((size_t)S + 1, 0);
    for (int x = 0; x < S; x++) nw.sp_first[(size_t)x + 1] = nw.sp_first[(size_t)x] + st.tiles.nsp[(size_t)x];
    FrameArgs &fr = nw.fr;
    {
        const double want = grel + 3.0 / 16777216.0;     // (+ 3u: see nbr_fast_prepare)
        float fg = (float)want;
        if ((double)fg < want) fg = nextafterf(fg, INFINITY);
        fr.guard_rel = fg;
    }
    fr.guard_abs = (float)(csum * (1.0 / 2147483648.0));
    fr.guard_abs16 = (float)(csum * (1.0 / 32768.0));      // (16-bit coordinates: one unit is 2^-16 of a cell vector; x2 margin as above)
    fr.sidx = nullptr;
    fr.sidx_stride = 0;
    return AMOF_OK;
}

// the tier takes the call: its tables (and whatever the caller added to nw.pk before) to the device in one copy
static int nbr_frame_commit(amof_ctx *ctx, NbrFrame &nw)
{
    const int i_cells = nw.pk.add(nw.cells.data(), nw.cells.size() * sizeof(NbrCell));
    const int i_sp = nw.pk.add(nw.sp_first.data(), nw.sp_first.size() * sizeof(int64_t));
    const int i_items = nw.pk.add(nw.items.data(), nw.items.size() * sizeof(FrameItem));
    AMOF_TRY(upload_pack(ctx, SLOT_AUX4, nw.pk));
    AMOF_TRY(ensure(ctx, SLOT_SPEC, sizeof(int32_t), &nw.d_qflag));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(nw.d_qflag, 0, sizeof(int32_t), ctx->stream));
    nw.fr.cells = nw.pk.ptr<NbrCell>(i_cells);
    nw.fr.sp_first = nw.d_spfirst = nw.pk.ptr<int64_t>(i_sp);
    nw.fr.items = nw.pk.ptr<FrameItem>(i_items);
    nw.fr.qflag = (int32_t *)nw.d_qflag;
    return AMOF_OK;
}

}  // namespace amof

using namespace amof;

extern "C" int amof_cn_count(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *sets,
                             int32_t n_sets, int64_t *sums, int32_t *per_atom)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(validate_traj(ctx, t, false));
    if (!cutoff || n_sets < 0 || (n_sets > 0 && (!sets || !sums))) return fail(ctx, AMOF_EINVAL, "NULL argument");
    const int S = t->n_species;
    for (int s = 0; s < n_sets; s++)
        if (sets[2 * s] < 0 || sets[2 * s] >= S || sets[2 * s + 1] < 0 || sets[2 * s + 1] >= S)
            return fail(ctx, AMOF_EINVAL, "set %d names a species out of range", s);
    if (n_sets == 0 || t->n_frames == 0) return AMOF_OK;
    NbrSetup st;
    AMOF_TRY(nbr_setup(ctx, t, cutoff, CN_TILE, st));
    std::vector<int4> work;
    for (int s = 0; s < n_sets; s++) {
        int A = sets[2 * s], B = sets[2 * s + 1];
        for (int k = 0; k < st.tiles.sp_ntiles[A]; k++)
            work.push_back(make_int4(s, st.tiles.sp_first_tile[A] + k, A, B));
    }
    const size_t F = (size_t)t->n_frames, N = (size_t)t->n_atoms;
    void *d_sums, *d_pa = nullptr;
    AMOF_TRY(ensure(ctx, SLOT_OUT0, F * n_sets * sizeof(int64_t), &d_sums));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_sums, 0, F * n_sets * sizeof(int64_t), ctx->stream));
    if (per_atom) {
        AMOF_TRY(ensure(ctx, SLOT_OUT1, F * n_sets * N * sizeof(int32_t), &d_pa));
        AMOF_HIP_TRY(ctx, hipMemsetAsync(d_pa, 0xFF, F * n_sets * N * sizeof(int32_t), ctx->stream));
    }
    NbrArgs &a = st.a;
    a.n_sets = n_sets;
    a.sums = (unsigned long long *)d_sums;
    a.per_atom = (int32_t *)d_pa;
    bool done = false;
    {
        // whole-frame-in-LDS tier: every set with a cutoff is one item (a zero-cutoff set with per-atom output keeps the
        // gather kernels: its centres still want their zeros)
        NbrFrame nw;
        double hmin[3];
        AMOF_TRY(nbr_frame_prepare(t, cutoff, st, nw, hmin));
        const bool tier_ok = nw.ok;
        bool ok16 = false;
        for (int pass = frame_first_pass(); pass < 3 && tier_ok; pass++) {
            // 16-byte records first; if a pair then needs a whole CU's LDS, everything again with the 8-byte ones; if that
            // does not bring two workgroups per CU either, the 16-byte ones stay -- unless they did not fit at all
            nw.ok = true; nw.items.clear(); nw.lds = 0; nw.compact = pass == 1;
            int64_t biggest = 0;
            for (int s2 = 0; s2 < n_sets && nw.ok; s2++) {
                const int A = sets[2 * s2], B = sets[2 * s2 + 1];
                const int64_t nA = st.tiles.nsp[A], nB = st.tiles.nsp[B];
                if (!(cutoff[A * S + B] > 0.0) || nA == 0 || nB == 0) { nw.ok = !per_atom; continue; }
                FrameItem it{};
                it.sa = A; it.sb = B; it.set = s2; it.reg_ab = it.reg_ba = -1;
                nw.ok = frame_item_grid(hmin, cutoff[A * S + B], A == B ? nA : nA + nB, 1, nB, it, nw.lds,
                                        nw.compact ? sizeof(uint2) : sizeof(uint4));
                if (nw.ok) nw.items.push_back(it);
                biggest = std::max(biggest, A == B ? nA : nA + nB);
            }
            if (pass == 0) ok16 = nw.ok;
            if (frame_pass_done(pass, nw.ok, nw.lds, ok16)) break;
        }
        // pairs too big for one workgroup (or AMOF_NBR_SLABS=1): every pair in z-slabs, the streaming kernels
        bool slabs = false;
        if (tier_ok && !frame_slabs_forbidden() && (frame_slabs_forced() || !nw.ok)) {
            bool zero_set = false;
            slabs = frame_slab_passes([&](size_t budget, bool thin, std::vector<FrameItem> &out, size_t &lds) {
                for (int s2 = 0; s2 < n_sets; s2++) {
                    const int A = sets[2 * s2], B = sets[2 * s2 + 1];
                    const int64_t nA = st.tiles.nsp[A], nB = st.tiles.nsp[B];
                    if (!(cutoff[A * S + B] > 0.0) || nA == 0 || nB == 0) { zero_set = true; continue; }
                    FrameItem it{};
                    it.sa = A; it.sb = B; it.set = s2; it.reg_ab = it.reg_ba = -1;
                    if (!frame_item_slabs(hmin, cutoff[A * S + B], nA, A == B ? 0 : nB, it, out, lds, budget, 0, 0, thin)) return false;
                }
                return true;
            }, nw.items, nw.lds);
            if (zero_set && per_atom) slabs = false;        // (a zero-cutoff set with per-atom output keeps the gather kernels)
            if (slabs) { nw.ok = true; nw.compact = false; }
            else if (frame_slabs_forced()) { nw.ok = false; }
        }
        if (nw.ok && !nw.items.empty()) {
            AMOF_TRY(nbr_frame_commit(ctx, nw));
            int64_t most = 0;
            for (const FrameItem &it : nw.items)
                most = std::max<int64_t>(most, st.tiles.nsp[it.sa] + (it.sa == it.sb ? 0 : st.tiles.nsp[it.sb]));
            int64_t launches = 0;
            int64_t FB = std::min<int64_t>(t->n_frames, 32768);
            if (nw.compact) {       // (atom, rank) of every sorted position: at most 256 MB a batch
                const size_t per_frame = nw.items.size() * (size_t)most * sizeof(uint2);
                FB = std::max<int64_t>(1, std::min<int64_t>(FB, (int64_t)(((size_t)256 << 20) / per_frame)));
                void *d_sidx;
                AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)FB * per_frame, &d_sidx));
                nw.fr.sidx = (uint2 *)d_sidx;
                nw.fr.sidx_stride = (int32_t)most;
            }
            if (slabs) AMOF_TRY(frame_slab_buffers(ctx, t, nw, FB));
            const int64_t FB0 = st.stage.lazy ? std::min<int64_t>(FB, 512) : FB;
            for (int64_t fb = 0, cur = FB0; fb < t->n_frames; fb += cur, cur = std::min<int64_t>(2 * cur, FB)) {
                const int64_t nfr = std::min<int64_t>(cur, t->n_frames - fb);
                AMOF_TRY(stager_need(st.stage, fb + nfr));
                nw.fr.f_base = (int32_t)fb;
                nw.fr.nf = (int32_t)nfr;
                if (launches == 0) timing_dom_begin(ctx, slabs ? "cn_frame_slabs" : "cn_frame");
                if (slabs) AMOF_TRY(frame_slab_quantize(ctx, t, a, nw, fb, nfr));
                const dim3 grid((unsigned)nw.items.size(), (unsigned)nfr);
                auto launch = [&](auto kern) -> hipError_t {
                    hipError_t e2 = allow_max_lds((const void *)kern);
                    if (e2 == hipSuccess) hipLaunchKernelGGL(kern, grid, dim3(NBRW_THREADS), nw.lds, ctx->stream, a, nw.fr);
                    return e2;
                };
                hipError_t e;
                if (slabs) e = nw.ortho ? launch(cn_frame_kernel<true, 0, false>) : launch(cn_frame_kernel<false, 0, false>);
                else if (most > 4 * NBRW_THREADS && nw.compact) e = nw.ortho ? launch(cn_frame_kernel<true, 8, true>) : launch(cn_frame_kernel<false, 8, true>);
                else if (most > 4 * NBRW_THREADS) e = nw.ortho ? launch(cn_frame_kernel<true, 8, false>) : launch(cn_frame_kernel<false, 8, false>);
                else if (nw.compact) e = nw.ortho ? launch(cn_frame_kernel<true, 4, true>) : launch(cn_frame_kernel<false, 4, true>);
                else e = nw.ortho ? launch(cn_frame_kernel<true, 4, false>) : launch(cn_frame_kernel<false, 4, false>);
                AMOF_HIP_TRY(ctx, e);
                AMOF_HIP_TRY(ctx, hipGetLastError());
                launches++;
            }
            timing_dom_end(ctx, launches);
            int32_t qflag = 0;
            AMOF_TRY(fetch(ctx, &qflag, nw.d_qflag, sizeof qflag));
            AMOF_HIP_TRY(ctx, sync_stream(ctx));
            if (qflag) {    // atoms absurdly far from the cell: the exact kernel answers (via the fast path's own check)
                AMOF_HIP_TRY(ctx, hipMemsetAsync(d_sums, 0, F * n_sets * sizeof(int64_t), ctx->stream));
                if (per_atom) AMOF_HIP_TRY(ctx, hipMemsetAsync(d_pa, 0xFF, F * n_sets * N * sizeof(int32_t), ctx->stream));
            } else {
                done = true;
            }
        } else if (nw.ok) {
            done = true;        // no set has a cutoff and both species: every count is zero
        }
    }
    NbrFast nf;
    unsigned long long centre_mask = 0ull;
    if (per_atom)
        for (int s2 = 0; s2 < n_sets; s2++)
            if (sets[2 * s2] < 64) centre_mask |= 1ull << sets[2 * s2];
    if (!done) AMOF_TRY(nbr_fast_prepare(ctx, t, cutoff, st, nf, centre_mask));
    if (nf.ok && !done) {
        std::vector<int4> fwork;
        for (int s2 = 0; s2 < n_sets; s2++) {
            int A = sets[2 * s2], B = sets[2 * s2 + 1];
            if (!(cutoff[A * S + B] > 0.0) && !per_atom) continue;      // counts stay zero (per_atom still wants its zeros)
            for (int64_t c0 = 0; c0 < st.tiles.nsp[A]; c0 += NBRF_TILE) fwork.push_back(make_int4(s2, (int)c0, A, B));
        }
        void *d_fwork;
        AMOF_TRY(upload(ctx, SLOT_AUX3, fwork.data(), fwork.size() * sizeof(int4), &d_fwork));
        nf.fa.a = a;
        nf.fa.a.work = (const int4 *)d_fwork;
        int64_t launches = 0;
        for (int64_t fb = 0, cur = nf.FB0; fb < t->n_frames && !fwork.empty(); fb += cur, cur = std::min<int64_t>(2 * cur, nf.FB)) {
            const int64_t nfr = std::min<int64_t>(cur, t->n_frames - fb);
            AMOF_TRY(stager_need(st.stage, fb + nfr));
            AMOF_TRY(nbr_fast_batch(ctx, t, st, nf, fb, nfr));
            unsigned chunks;
            pick_chunks(nfr, fwork.size(), nf.fa.a.frames_per_chunk, chunks);
            dim3 grid((unsigned)fwork.size(), chunks);
            if (launches == 0) timing_dom_begin(ctx, nf.cell ? "cn_cell" : "cn_fast");
            if (nf.cell && nf.ortho) hipLaunchKernelGGL((cn_fast_kernel<true, true>), grid, dim3(NBRF_TILE), 0, ctx->stream, nf.fa);
            else if (nf.cell) hipLaunchKernelGGL((cn_fast_kernel<false, true>), grid, dim3(NBRF_TILE), 0, ctx->stream, nf.fa);
            else if (nf.ortho) hipLaunchKernelGGL((cn_fast_kernel<true, false>), grid, dim3(NBRF_TILE), 0, ctx->stream, nf.fa);
            else hipLaunchKernelGGL((cn_fast_kernel<false, false>), grid, dim3(NBRF_TILE), 0, ctx->stream, nf.fa);
            AMOF_HIP_TRY(ctx, hipGetLastError());
            launches++;
        }
        timing_dom_end(ctx, launches);
        int32_t qflag = 0;
        AMOF_TRY(fetch(ctx, &qflag, nf.d_qflag, sizeof qflag));
        AMOF_HIP_TRY(ctx, sync_stream(ctx));
        if (qflag) {   // atoms absurdly far from the cell: redo with the exact kernel
            AMOF_HIP_TRY(ctx, hipMemsetAsync(d_sums, 0, F * n_sets * sizeof(int64_t), ctx->stream));
        } else {
            done = true;
        }
    }
    AMOF_TRY(stager_need(st.stage, t->n_frames));   // (no-op unless the fast path was skipped)
    if (!done && !work.empty()) {
        void *d_work;       // (the exact kernel's work list: uploaded only when it runs)
        AMOF_TRY(upload(ctx, SLOT_PAIRS, work.data(), work.size() * sizeof(int4), &d_work));
        a.work = (const int4 *)d_work;
        unsigned chunks;
        pick_chunks(t->n_frames, work.size(), a.frames_per_chunk, chunks);
        dim3 grid((unsigned)work.size(), chunks);
        const bool extra = st.max_img > 0, ortho = st.geom.all_ortho;
        timing_dom_begin(ctx, "cn_exact");
        if (ortho && !extra) hipLaunchKernelGGL((cn_kernel<true, false>), grid, dim3(CN_TILE), 0, ctx->stream, a);
        else if (ortho && extra) hipLaunchKernelGGL((cn_kernel<true, true>), grid, dim3(CN_TILE), 0, ctx->stream, a);
        else if (!ortho && !extra) hipLaunchKernelGGL((cn_kernel<false, false>), grid, dim3(CN_TILE), 0, ctx->stream, a);
        else hipLaunchKernelGGL((cn_kernel<false, true>), grid, dim3(CN_TILE), 0, ctx->stream, a);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        timing_dom_end(ctx, 1);
    }
    timing_end(ctx);
    AMOF_TRY(fetch(ctx, sums, d_sums, F * n_sets * sizeof(int64_t)));
    if (per_atom)
        AMOF_TRY(fetch(ctx, per_atom, d_pa, F * n_sets * N * sizeof(int32_t)));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

static int bad_run(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *triples,
                   int32_t T, const double *edges, int32_t nb, unsigned long long *hist_dev,
                   unsigned long long *nang_dev, int32_t cn_max = 0)
{
    NbrSetup st;
    AMOF_TRY(nbr_setup(ctx, t, cutoff, BAD_TILE, st));
    std::vector<int4> work;
    for (int k = 0; k < T; k++) {
        int A = triples[2 * k], B = triples[2 * k + 1];
        for (int tl = 0; tl < (int)st.tiles.tiles.size(); tl++)
            if (A < 0 || st.tiles.tiles[tl].species == A) work.push_back(make_int4(k, tl, A, B));
    }
    void *d_edges, *d_flags;
    AMOF_TRY(upload(ctx, SLOT_AUX3, edges, (size_t)(nb + 1) * sizeof(double), &d_edges));
    AMOF_TRY(ensure(ctx, SLOT_FLAGS, 4 * sizeof(int32_t), &d_flags));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, 4 * sizeof(int32_t), ctx->stream));
    NbrArgs &a = st.a;
    a.edges = (const double *)d_edges;
    a.nb = nb;
    a.edge_step = 0.0;
    if (edges[0] == 0.0 && !getenv("AMOF_BAD_EDGE_TABLE")) {
        volatile double step = edges[1];
        bool uniform = true;
        for (int k = 0; k <= nb && uniform; k++) {
            volatile double e = (double)k * step;      // (one IEEE multiplication, as numpy and the kernels do it)
            uniform = e == edges[k];
        }
        if (uniform) a.edge_step = step;
    }
    a.flags = (int32_t *)d_flags;
    a.cn_max = cn_max;
    a.nbuf = nullptr;
    a.ncap = 0;
    a.count_only = 0;
    a.global_hist = nb > AMOF_MAX_LDS_BINS - 16384 ? 1 : 0;      // (the reference has no limit on the bin count)
    const size_t lds_bins = a.global_hist ? 0 : (size_t)nb;
    const size_t KC = (size_t)(cn_max > 0 ? cn_max + 1 : 1);   // histogram slots per triple
    // every pass accumulates into scratch; the caller's buffers only ever receive a complete, valid result
    const size_t hs_words = (size_t)T * KC * nb + (size_t)T * KC;
    void *d_hs, *d_ns;
    AMOF_TRY(ensure(ctx, SLOT_AUX7, hs_words * sizeof(unsigned long long), &d_hs));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_hs, 0, hs_words * sizeof(unsigned long long), ctx->stream));
    d_ns = (unsigned long long *)d_hs + (size_t)T * KC * nb;
    a.hist = (unsigned long long *)d_hs;
    a.n_angles = (unsigned long long *)d_ns;
    auto read_flags = [&](int32_t (&fl)[4]) -> int {
        AMOF_TRY(fetch(ctx, fl, d_flags, sizeof fl));
        AMOF_HIP_TRY(ctx, sync_stream(ctx));
        return AMOF_OK;
    };
    auto clear_scratch = [&]() -> int {
        AMOF_HIP_TRY(ctx, hipMemsetAsync(d_hs, 0, hs_words * sizeof(unsigned long long), ctx->stream));
        AMOF_HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, 4 * sizeof(int32_t), ctx->stream));
        return AMOF_OK;
    };
    bool done = false, overflow = false;
    int32_t flags[4] = {0, 0, 0, 0};
    {
        // ---- whole-frame-in-LDS tier: one sort + LDS search per species pair into neighbour rows, every triple from the rows
        const int S = t->n_species;
        NbrFrame nw;
        double hmin[3];
        AMOF_TRY(nbr_frame_prepare(t, cutoff, st, nw, hmin));
        if (getenv("AMOF_BAD_NOTRANSPOSE")) nw.ok = false;      // (names the two-search gather kernels)
        auto live_pair = [&](int x, int y) { return cutoff[x * S + y] > 0.0 && st.tiles.nsp[x] > 0 && st.tiles.nsp[y] > 0; };
        std::vector<char> needed((size_t)S * S, 0);
        std::vector<int4> awork;
        for (int k = 0; k < T && nw.ok; k++) {
            const int A = triples[2 * k], B = triples[2 * k + 1];
            for (int sa = 0; sa < S; sa++) {
                if (!(A < 0 || sa == A)) continue;
                bool live = false;
                for (int sb = 0; sb < S; sb++)
                    if ((B < 0 || sb == B) && live_pair(sa, sb)) { needed[(size_t)sa * S + sb] = 1; live = true; }
                if (live) awork.push_back(make_int4(k, sa, B, 0));
            }
        }
        std::vector<int32_t> region_of((size_t)S * S, -1);
        int64_t R = 0;
        for (int x = 0; x < S * S && nw.ok; x++)
            if (needed[(size_t)x]) { region_of[(size_t)x] = (int32_t)R; R += st.tiles.nsp[(size_t)(x / S)]; }
        int64_t most = 0;
        const bool tier_ok = nw.ok;
        bool ok16 = false;
        for (int pass = frame_first_pass(); pass < 3 && tier_ok; pass++) {
            // 16-byte records first; if a pair then needs a whole CU's LDS, everything again with the 8-byte ones; if that
            // does not bring two workgroups per CU either, the 16-byte ones stay -- unless they did not fit at all
            nw.ok = true; nw.items.clear(); nw.lds = 0; nw.compact = pass == 1; most = 0;
            for (int x = 0; x < S && nw.ok; x++)
                for (int y = x; y < S && nw.ok; y++) {
                    if (!needed[(size_t)x * S + y] && !needed[(size_t)y * S + x]) continue;
                    FrameItem it{};
                    it.sa = st.tiles.nsp[y] < st.tiles.nsp[x] ? y : x;      // the species with fewer atoms searches
                    it.sb = it.sa == x ? y : x;
                    it.set = 0;
                    it.reg_ab = region_of[(size_t)it.sa * S + it.sb];
                    it.reg_ba = x == y ? -1 : region_of[(size_t)it.sb * S + it.sa];
                    const int64_t n = st.tiles.nsp[x] + (x == y ? 0 : st.tiles.nsp[y]);
                    size_t lds_it = 0;
                    nw.ok = frame_item_grid(hmin, cutoff[x * S + y], n, 1, st.tiles.nsp[it.sb], it, lds_it,
                                            nw.compact ? sizeof(uint2) : sizeof(uint4),
                                            (size_t)((n + 3) / 4) * 4 + (size_t)NBRW_HITS * sizeof(uint32_t));
                    nw.lds = std::max(nw.lds, lds_it);
                    most = std::max(most, n);
                    if (nw.ok) nw.items.push_back(it);
                }
            if (pass == 0) ok16 = nw.ok;
            if (frame_pass_done(pass, nw.ok, nw.lds, ok16)) break;
        }
        // pairs too big for one workgroup (or AMOF_NBR_SLABS=1): every pair in z-slabs, the streaming kernels
        bool slabs = false, shared_rows = false;
        if (tier_ok && !frame_slabs_forbidden() && (frame_slabs_forced() || !nw.ok)) {
            slabs = frame_slab_passes([&](size_t budget, bool thin, std::vector<FrameItem> &out, size_t &lds) {
                for (int x = 0; x < S; x++)
                    for (int y = x; y < S; y++) {
                        if (!needed[(size_t)x * S + y] && !needed[(size_t)y * S + x]) continue;
                        FrameItem it{};
                        it.sa = st.tiles.nsp[y] < st.tiles.nsp[x] ? y : x;      // the species with fewer atoms searches
                        it.sb = it.sa == x ? y : x;
                        it.set = 0;
                        it.reg_ab = region_of[(size_t)it.sa * S + it.sb];
                        it.reg_ba = x == y ? -1 : region_of[(size_t)it.sb * S + it.sa];
                        if (!frame_item_slabs(hmin, cutoff[x * S + y], st.tiles.nsp[it.sa], x == y ? 0 : st.tiles.nsp[it.sb], it, out, lds,
                                              budget, (size_t)NBRW_HITS * sizeof(uint32_t) + 4, 1, thin)) return false;
                    }
                return true;
            }, nw.items, nw.lds);
            if (slabs) {
                nw.ok = true; nw.compact = false;
                for (const FrameItem &it : nw.items) shared_rows = shared_rows || (it.cn != it.nzg && it.reg_ba >= 0);
            } else if (frame_slabs_forced()) {
                nw.ok = false;
            }
        }
        if (nw.ok && !awork.empty() && R > 0 && R < (1ll << 30) && t->n_frames > 0) {
            // merged angle passes: one per centre species, every angle computed once (histograms in LDS; BadByCn keys, more
            // bins than LDS holds, a centre species with more than three partner species or a triple named twice keep
            // the pass per triple)
            std::vector<MergedItem> merged;
            size_t lds_merged = 0;
            if (cn_max == 0 && !a.global_hist && !getenv("AMOF_BAD_NOMERGE")) {
                auto find_triple = [&](int A, int B, bool &twice) {
                    int found = -1;
                    for (int k = 0; k < T; k++)
                        if (triples[2 * k] == A && triples[2 * k + 1] == B) { if (found >= 0) twice = true; else found = k; }
                    return found;
                };
                bool usable = true, twice = false;
                for (int sa = 0; sa < S && usable; sa++) {
                    MergedItem mi{};
                    mi.sa = sa;
                    for (int q = 0; q < 3; q++) { mi.reg[q] = 0; mi.tb[q][0] = mi.tb[q][1] = -1; }
                    for (int sb = 0; sb < S && usable; sb++) {
                        if (region_of[(size_t)sa * S + sb] < 0) continue;
                        if (mi.n_reg == 3) { usable = false; break; }
                        mi.reg[mi.n_reg] = region_of[(size_t)sa * S + sb];
                        mi.tb[mi.n_reg][0] = find_triple(sa, sb, twice);
                        mi.tb[mi.n_reg][1] = find_triple(-1, sb, twice);
                        mi.n_reg++;
                    }
                    mi.tx[0] = find_triple(sa, -1, twice);
                    mi.tx[1] = find_triple(-1, -1, twice);
                    if (mi.n_reg == 0) continue;
                    bool any = mi.tx[0] >= 0 || mi.tx[1] >= 0;
                    for (int q = 0; q < mi.n_reg; q++) any = any || mi.tb[q][0] >= 0 || mi.tb[q][1] >= 0;
                    if (!any) continue;
                    lds_merged = std::max(lds_merged, (size_t)(mi.n_reg + (mi.n_reg > 1 ? 1 : 0)) * lds_bins * sizeof(unsigned));
                    merged.push_back(mi);
                }
                // (worth it when it saves passes: two triples over one species pair are two passes either way, and the pass
                //  per triple keeps a centre's vectors in registers)
                if (!usable || twice || lds_merged > 60 * 1024 || merged.size() >= awork.size()) merged.clear();
            }
            // one small table: angle work | region_of[S*S] | inv_rank[N]; the items beside it
            std::vector<int32_t> tab(4 * awork.size() + (size_t)S * S + (size_t)t->n_atoms);
            memcpy(tab.data(), awork.data(), awork.size() * sizeof(int4));
            memcpy(&tab[4 * awork.size()], region_of.data(), region_of.size() * sizeof(int32_t));
            for (int64_t x = 0; x < t->n_atoms; x++) {
                const int32_t atom = st.tiles.perm[(size_t)x];
                tab[4 * awork.size() + (size_t)S * S + (size_t)atom] = (int32_t)(x - nw.sp_first[(size_t)t->species[atom]]);
            }
            void *d_lists;
            const int i_tab = nw.pk.add(tab.data(), tab.size() * sizeof(int32_t));
            const int i_merged = nw.pk.add(merged.data(), merged.size() * sizeof(MergedItem));
            AMOF_TRY(nbr_frame_commit(ctx, nw));
            const int32_t *d_tab = nw.pk.ptr<int32_t>(i_tab);
            const size_t per_frame = (size_t)R * (sizeof(uint32_t) + (size_t)NBRL_CAP * NBRL_EW * sizeof(double));
            size_t rows_budget = (size_t)4 << 30;                          // <= 4 GiB of rows
            if (const char *mb = getenv("AMOF_BAD_ROWS_MB")) rows_budget = (size_t)std::max(1, atoi(mb)) << 20;     // tests: many batches
            int64_t FB = std::max<int64_t>(1, (int64_t)rows_budget / (int64_t)per_frame);
            FB = std::min<int64_t>(FB, std::max<int64_t>(1, 0x7fffff00ll / std::max<int64_t>(1, t->n_atoms)));    // flat (frame, centre) index
            FB = std::min<int64_t>(std::min<int64_t>(FB, 32768), t->n_frames);
            size_t sidx_per_frame = 0;
            if (nw.compact) {       // (atom, rank) of every sorted position: at most 256 MB a batch
                sidx_per_frame = nw.items.size() * (size_t)most * sizeof(uint2);
                FB = std::max<int64_t>(1, std::min<int64_t>(FB, (int64_t)(((size_t)256 << 20) / sidx_per_frame)));
            }
            // (a device short of memory gets smaller batches, not an error)
            for (;;) {
                const int rc_rows = ensure(ctx, SLOT_AUX9, (size_t)FB * per_frame, &d_lists);
                if (rc_rows == AMOF_OK) break;
                if (rc_rows != AMOF_ENOMEM || FB == 1) return rc_rows;
                FB = std::max<int64_t>(1, FB / 4);
            }
            if (slabs) AMOF_TRY(frame_slab_buffers(ctx, t, nw, FB));
            const int64_t FB0 = st.stage.lazy ? std::min<int64_t>(FB, 512) : FB;
            if (nw.compact) {
                void *d_sidx;
                AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)FB * sidx_per_frame, &d_sidx));
                nw.fr.sidx = (uint2 *)d_sidx;
                nw.fr.sidx_stride = (int32_t)most;
            }
            NbrListArgs la;
            const int4 *d_aw = (const int4 *)d_tab;
            la.region_of = (const int32_t *)d_tab + 4 * awork.size();
            la.inv_rank = la.region_of + (size_t)S * S;
            la.rows = (double *)d_lists;                                   // (doubles first: 8-byte aligned)
            la.count = (uint32_t *)(la.rows + (size_t)FB * R * NBRL_CAP * NBRL_EW);
            la.R = (int32_t)R;
            la.plane = (size_t)FB * (size_t)R;
            const size_t lds_rows = lds_bins * sizeof(unsigned);
            int64_t launches = 0;
            for (int64_t fb = 0, cur = FB0; fb < t->n_frames; fb += cur, cur = std::min<int64_t>(2 * cur, FB)) {
                const int64_t nfr = std::min<int64_t>(cur, t->n_frames - fb);
                AMOF_TRY(stager_need(st.stage, fb + nfr));
                nw.fr.f_base = (int32_t)fb;
                nw.fr.nf = (int32_t)nfr;
                if (launches == 0) timing_dom_begin(ctx, slabs ? "bad_frame_slabs" : "bad_frame");
                if (slabs) AMOF_TRY(frame_slab_quantize(ctx, t, a, nw, fb, nfr));
                if (shared_rows)    // (partners of several slabs claim their rows' slots with global atomics: the counts start at zero)
                    AMOF_HIP_TRY(ctx, hipMemsetAsync(la.count, 0, (size_t)FB * (size_t)R * sizeof(uint32_t), ctx->stream));
                const dim3 sgrid((unsigned)nw.items.size(), (unsigned)nfr);
                auto launch = [&](auto kern) -> hipError_t {
                    hipError_t e2 = allow_max_lds((const void *)kern);
                    if (e2 == hipSuccess) hipLaunchKernelGGL(kern, sgrid, dim3(NBRW_THREADS), nw.lds, ctx->stream, a, nw.fr, la);
                    return e2;
                };
                hipError_t e;
                if (slabs) e = nw.ortho ? launch(lists_frame_kernel<true, 0, false>) : launch(lists_frame_kernel<false, 0, false>);
                else if (most > 4 * NBRW_THREADS && nw.compact) e = nw.ortho ? launch(lists_frame_kernel<true, 8, true>) : launch(lists_frame_kernel<false, 8, true>);
                else if (most > 4 * NBRW_THREADS) e = nw.ortho ? launch(lists_frame_kernel<true, 8, false>) : launch(lists_frame_kernel<false, 8, false>);
                else if (nw.compact) e = nw.ortho ? launch(lists_frame_kernel<true, 4, true>) : launch(lists_frame_kernel<false, 4, true>);
                else e = nw.ortho ? launch(lists_frame_kernel<true, 4, false>) : launch(lists_frame_kernel<false, 4, false>);
                AMOF_HIP_TRY(ctx, e);
                AMOF_HIP_TRY(ctx, hipGetLastError());
                if (!merged.empty()) {
                    // one pass per centre species: about 1024 workgroups of 512 lanes in all
                    int64_t widest_m = 0;
                    for (const MergedItem &mi : merged)
                        widest_m = std::max<int64_t>(widest_m, (nfr * st.tiles.nsp[(size_t)mi.sa] + NBRM_THREADS - 1) / NBRM_THREADS);
                    const int64_t gm = std::max<int64_t>(1, std::min<int64_t>(widest_m, (1024 + (int64_t)merged.size() - 1) / (int64_t)merged.size()));
                    const dim3 mgrid((unsigned)gm, (unsigned)merged.size());
                    const MergedItem *d_merged = nw.pk.ptr<MergedItem>(i_merged);
                    if (nw.ortho) {
                        e = allow_max_lds((const void *)bad_rows_merged_kernel<true>);
                        if (e == hipSuccess) hipLaunchKernelGGL(bad_rows_merged_kernel<true>, mgrid, dim3(NBRM_THREADS), lds_merged, ctx->stream, a, la,
                                                                d_merged, nw.d_spfirst, (int)nfr);
                    } else {
                        e = allow_max_lds((const void *)bad_rows_merged_kernel<false>);
                        if (e == hipSuccess) hipLaunchKernelGGL(bad_rows_merged_kernel<false>, mgrid, dim3(NBRM_THREADS), lds_merged, ctx->stream, a, la,
                                                                d_merged, nw.d_spfirst, (int)nfr);
                    }
                    AMOF_HIP_TRY(ctx, e);
                    AMOF_HIP_TRY(ctx, hipGetLastError());
                    launches++;
                    continue;
                }
                // angle kernel: about eight workgroups per CU in all work items together, each striding over the tiles of its own
                int64_t widest = 0;
                for (const int4 &w : awork) widest = std::max<int64_t>(widest, (nfr * st.tiles.nsp[(size_t)w.y] + NBRF_TILE - 1) / NBRF_TILE);
                // (every workgroup ends with one global atomic per non-empty bin of its histogram, all on the same few
                //  addresses: 16 384 workgroups spent more time there than on the angles)
                const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(widest, (2048 + (int64_t)awork.size() - 1) / (int64_t)awork.size()));
                const dim3 agrid((unsigned)gx, (unsigned)awork.size());
                if (nw.ortho) {
                    e = allow_max_lds((const void *)bad_rows_kernel<true>);
                    if (e == hipSuccess) hipLaunchKernelGGL(bad_rows_kernel<true>, agrid, dim3(NBRF_TILE), lds_rows, ctx->stream, a, la, d_aw,
                                                            nw.d_spfirst, (int)nfr);
                } else {
                    e = allow_max_lds((const void *)bad_rows_kernel<false>);
                    if (e == hipSuccess) hipLaunchKernelGGL(bad_rows_kernel<false>, agrid, dim3(NBRF_TILE), lds_rows, ctx->stream, a, la, d_aw,
                                                            nw.d_spfirst, (int)nfr);
                }
                AMOF_HIP_TRY(ctx, e);
                AMOF_HIP_TRY(ctx, hipGetLastError());
                launches++;
            }
            timing_dom_end(ctx, launches);
#ifdef NBR_PHASE_STAMPS
            {
                unsigned long long ph[16][5];
                hipDeviceSynchronize();
                hipMemcpyFromSymbol(ph, HIP_SYMBOL(nbr_phase_ticks), sizeof ph);
                for (size_t e = 0; e < nw.items.size() && e < 16; e++)
                    if (ph[e][4])
                        fprintf(stderr, "lists entry %zu (species %d+%d, layers %d of %d, grid %dx%dx%d): us per workgroup: sort %.1f search + unit vectors (wave 0) %.1f (-) %.1f counts %.1f\n",
                                e, nw.items[e].sa, nw.items[e].sb, nw.items[e].cn, nw.items[e].nzg, nw.items[e].nx, nw.items[e].ny, nw.items[e].nz,
                                0.01 * ph[e][0] / ph[e][4], 0.01 * ph[e][1] / ph[e][4], 0.01 * ph[e][2] / ph[e][4], 0.01 * ph[e][3] / ph[e][4]);
                memset(ph, 0, sizeof ph);
                hipMemcpyToSymbol(HIP_SYMBOL(nbr_phase_ticks), ph, sizeof ph);
            }
#endif
            int32_t qflag = 0;
            AMOF_TRY(fetch(ctx, &qflag, nw.d_qflag, sizeof qflag));
            AMOF_TRY(read_flags(flags));
            if (flags[0]) return fail(ctx, AMOF_EANGLE, "Undefined angle");
            if (qflag || flags[1]) {
                // atoms absurdly far from the cell, or a centre with more than NBRL_CAP neighbours: the gather / exact kernels
                AMOF_TRY(clear_scratch());
            } else {
                done = true;
            }
        } else if (nw.ok && awork.empty()) {
            done = true;            // no triple has a centre with a cutoff to any of its partners: no angle
        }
    }
    NbrFast nf;
    if (!done) AMOF_TRY(nbr_fast_prepare(ctx, t, cutoff, st, nf));
    if (nf.ok && !done && t->n_frames > 0) {
        const int S = t->n_species;
        // transposed lists: of two triples B-A-B / A-B-A over one species pair, only the side with fewer centres searches
        std::vector<int32_t> derived_from((size_t)T, -1), tr_off((size_t)T, -1);
        std::vector<TrDerived> der;
        int32_t tr_total = 0;
        if (!getenv("AMOF_BAD_NOTRANSPOSE")) {
            for (int k = 0; k < T; k++) {
                const int A = triples[2 * k], B = triples[2 * k + 1];
                if (A < 0 || B < 0 || A == B || !(cutoff[A * S + B] > 0.0) || !(st.tiles.nsp[A] > st.tiles.nsp[B])) continue;
                for (int k2 = 0; k2 < T && derived_from[(size_t)k] < 0; k2++)
                    if (triples[2 * k2] == B && triples[2 * k2 + 1] == A && tr_off[(size_t)k2] < 0) {
                        derived_from[(size_t)k] = k2;
                        tr_off[(size_t)k2] = tr_total;
                        der.push_back(TrDerived{k, A, tr_total, (int32_t)st.tiles.nsp[A]});
                        tr_total += (int32_t)st.tiles.nsp[A];
                    }
            }
        }
        std::vector<int4> fwork;
        for (int k = 0; k < T; k++) {
            int A = triples[2 * k], B = triples[2 * k + 1];
            if (derived_from[(size_t)k] >= 0) continue;      // (its angles come from the lists of the triple searched from the other side)
            for (int sa = 0; sa < S; sa++) {
                if (!(A < 0 || sa == A)) continue;
                // centres of a species that has no cutoff with any partner species of this triple find nothing
                bool live = false;
                for (int sb = 0; sb < S; sb++)
                    if ((B < 0 || sb == B) && cutoff[sa * S + sb] > 0.0 && st.tiles.nsp[sb] > 0) live = true;
                if (!live) continue;
                for (int64_t c0 = 0; c0 < st.tiles.nsp[sa]; c0 += NBRF_TILE) fwork.push_back(make_int4(k, (int)c0, sa, B));
            }
        }
        void *d_fwork;
        AMOF_TRY(upload(ctx, SLOT_AUX6, fwork.data(), fwork.size() * sizeof(int4), &d_fwork));
        void *d_trtab = nullptr, *d_trlists = nullptr;
        int n_twork = 0;
        if (tr_total > 0) {
            // one small table: tr_off[T] | inv_rank[N] | derived records | their work list; the lists shrink the frame
            // batch if they must
            std::vector<int2> twork;
            for (size_t dd = 0; dd < der.size(); dd++)
                for (int32_t c0 = 0; c0 < der[dd].count; c0 += 256) twork.push_back(make_int2((int)dd, c0));
            n_twork = (int)twork.size();
            std::vector<int32_t> tab((size_t)T + (size_t)t->n_atoms + 4 * der.size() + 2 * twork.size());
            for (int k = 0; k < T; k++) tab[(size_t)k] = tr_off[(size_t)k];
            for (int64_t x = 0; x < t->n_atoms; x++) {
                const int32_t atom = st.tiles.perm[(size_t)x];
                tab[(size_t)T + (size_t)atom] = (int32_t)(x - nf.sp_first[t->species[atom]]);
            }
            memcpy(&tab[(size_t)T + (size_t)t->n_atoms], der.data(), der.size() * sizeof(TrDerived));
            memcpy(&tab[(size_t)T + (size_t)t->n_atoms + 4 * der.size()], twork.data(), twork.size() * sizeof(int2));
            AMOF_TRY(upload(ctx, SLOT_AUX8, tab.data(), tab.size() * sizeof(int32_t), &d_trtab));
            const size_t per_frame = (size_t)tr_total * (1 + TR_CAP) * sizeof(uint32_t);
            const int64_t fit = std::max<int64_t>(1, (int64_t)((size_t)1 << 30) / (int64_t)per_frame);
            nf.FB = std::min<int64_t>(nf.FB, fit);
            nf.FB0 = std::min<int64_t>(nf.FB0, nf.FB);
            AMOF_TRY(ensure(ctx, SLOT_AUX9, (size_t)nf.FB * per_frame, &d_trlists));
        }
        nf.fa.a = a;
        nf.fa.a.work = (const int4 *)d_fwork;
        nf.fa.a.tr_total = tr_total;
        if (tr_total > 0) {
            nf.fa.a.tr_off = (const int32_t *)d_trtab;
            nf.fa.a.inv_rank = (const int32_t *)d_trtab + T;
            nf.fa.a.tcount = (uint32_t *)d_trlists;
            nf.fa.a.tlist = (uint32_t *)d_trlists + (size_t)nf.FB * tr_total;
        }
        size_t lds = 3 * (size_t)NBRF_UVCAP * sizeof(double) +
                     (size_t)NBRF_NLIST * NBRF_TILE * sizeof(uint32_t) + NBRF_TILE * sizeof(uint32_t) +
                     (NBRF_TILE + 4) * sizeof(int) + NBRF_UVCAP * sizeof(unsigned short) + lds_bins * sizeof(unsigned);
        int64_t launches = 0;
        for (int64_t fb = 0, cur = nf.FB0; fb < t->n_frames && !fwork.empty(); fb += cur, cur = std::min<int64_t>(2 * cur, nf.FB)) {
            const int64_t nfr = std::min<int64_t>(cur, t->n_frames - fb);
            AMOF_TRY(stager_need(st.stage, fb + nfr));
            AMOF_TRY(nbr_fast_batch(ctx, t, st, nf, fb, nfr));
            if (tr_total > 0)
                AMOF_HIP_TRY(ctx, hipMemsetAsync(nf.fa.a.tcount, 0, (size_t)nfr * tr_total * sizeof(uint32_t), ctx->stream));
            unsigned chunks;
            pick_chunks(nfr, fwork.size(), nf.fa.a.frames_per_chunk, chunks);
            dim3 grid((unsigned)fwork.size(), chunks);
            if (launches == 0) timing_dom_begin(ctx, nf.cell ? "bad_cell" : "bad_fast");
            auto launch = [&](auto kern) -> hipError_t {
                hipError_t e2 = allow_max_lds((const void *)kern);
                if (e2 == hipSuccess) hipLaunchKernelGGL(kern, grid, dim3(NBRF_TILE), lds, ctx->stream, nf.fa);
                return e2;
            };
            hipError_t e;
            if (nf.cell && nf.ortho) e = launch(bad_fast_kernel<true, true>);
            else if (nf.cell) e = launch(bad_fast_kernel<false, true>);
            else if (nf.ortho) e = launch(bad_fast_kernel<true, false>);
            else e = launch(bad_fast_kernel<false, false>);
            AMOF_HIP_TRY(ctx, e);
            AMOF_HIP_TRY(ctx, hipGetLastError());
            if (tr_total > 0) {     // the angles of the triples that were not searched, from the lists just written
                const TrDerived *d_der = reinterpret_cast<const TrDerived *>((const int32_t *)d_trtab + T + t->n_atoms);
                const int2 *d_twork = reinterpret_cast<const int2 *>((const int32_t *)d_trtab + T + t->n_atoms + 4 * der.size());
                NbrFastArgs ta = nf.fa;
                unsigned tchunks;
                pick_chunks(nfr, (size_t)n_twork, ta.a.frames_per_chunk, tchunks);
                dim3 tgrid((unsigned)n_twork, tchunks);
                const size_t tlds = lds_bins * sizeof(unsigned);
                hipError_t e3;
                if (nf.ortho) {
                    e3 = allow_max_lds((const void *)bad_transposed_kernel<true>);
                    if (e3 == hipSuccess) hipLaunchKernelGGL(bad_transposed_kernel<true>, tgrid, dim3(256), tlds, ctx->stream, ta, d_der, d_twork);
                } else {
                    e3 = allow_max_lds((const void *)bad_transposed_kernel<false>);
                    if (e3 == hipSuccess) hipLaunchKernelGGL(bad_transposed_kernel<false>, tgrid, dim3(256), tlds, ctx->stream, ta, d_der, d_twork);
                }
                AMOF_HIP_TRY(ctx, e3);
                AMOF_HIP_TRY(ctx, hipGetLastError());
            }
            launches++;
        }
        timing_dom_end(ctx, launches);
        int32_t qflag = 0;
        AMOF_TRY(fetch(ctx, &qflag, nf.d_qflag, sizeof qflag));
        AMOF_TRY(read_flags(flags));
        if (flags[0]) return fail(ctx, AMOF_EANGLE, "Undefined angle");
        if (qflag) {           // atoms absurdly far from the cell: redo with the exact kernel
            AMOF_TRY(clear_scratch());
        } else if (flags[1]) {
            // a centre has more than NBRF_NLIST neighbours: the exact kernel with its AMOF_MAX_NEIGHBOURS-deep LDS lists
            // comes next (dense systems with 17..32 neighbours stay in LDS); only if that overflows too, the big-list pass
            AMOF_TRY(clear_scratch());
        } else {
            done = true;
        }
    }
    AMOF_TRY(stager_need(st.stage, t->n_frames));   // (no-op unless the fast path was skipped)
    const bool extra = st.max_img > 0, ortho = st.geom.all_ortho;
    const size_t lds_exact = (size_t)(3 * AMOF_MAX_NEIGHBOURS * BAD_TILE + 3 * BAD_TILE) * sizeof(double) +
                             BAD_TILE * sizeof(int) + lds_bins * sizeof(unsigned);
    if (!done && !work.empty() && t->n_frames > 0) {      // (the exact kernels' work list: uploaded only when they run)
        void *d_work;
        AMOF_TRY(upload(ctx, SLOT_PAIRS, work.data(), work.size() * sizeof(int4), &d_work));
        a.work = (const int4 *)d_work;
    }
    auto launch_exact = [&](dim3 grid) -> hipError_t {
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e = allow_max_lds((const void *)kern);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(BAD_TILE), lds_exact, ctx->stream, a);
            return hipGetLastError();
        };
        if (ortho && !extra) return launch(bad_kernel<true, false>);
        if (ortho && extra) return launch(bad_kernel<true, true>);
        if (!ortho && !extra) return launch(bad_kernel<false, false>);
        return launch(bad_kernel<false, true>);
    };
    if (!done && !overflow && !work.empty() && t->n_frames > 0) {
        unsigned chunks;
        pick_chunks(t->n_frames, work.size(), a.frames_per_chunk, chunks);
        timing_dom_begin(ctx, "bad_exact");
        AMOF_HIP_TRY(ctx, launch_exact(dim3((unsigned)work.size(), chunks)));
        timing_dom_end(ctx, 1);
        AMOF_TRY(read_flags(flags));
        if (flags[0]) return fail(ctx, AMOF_EANGLE, "Undefined angle");
        if (flags[1]) {
            overflow = true;
            AMOF_TRY(clear_scratch());
        }
    }
    if (overflow && !work.empty() && t->n_frames > 0) {
        // Big-list pass.  The reference has no limit on the neighbours of a centre (amof/bad.py:87-100): find the
        // largest neighbour count with a counting pass of the exact kernel, then run that kernel once more with its
        // per-centre lists of unit vectors in global scratch, [workgroup][3][cap][BAD_TILE] doubles.
        unsigned chunks;
        pick_chunks(t->n_frames, work.size(), a.frames_per_chunk, chunks);
        a.count_only = 1;
        AMOF_HIP_TRY(ctx, launch_exact(dim3((unsigned)work.size(), chunks)));
        AMOF_TRY(read_flags(flags));
        a.count_only = 0;
        const int64_t cap = std::max<int64_t>(flags[2], 1);
        const size_t per_wg = (size_t)3 * (size_t)cap * BAD_TILE * sizeof(double);
        const size_t budget = (size_t)2 << 30;
        // fewer, longer frame chunks until the scratch of all workgroups fits the budget
        int64_t max_wg = std::max<int64_t>(1, (int64_t)(budget / per_wg));
        int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>(chunks, max_wg / (int64_t)work.size()));
        a.frames_per_chunk = (int32_t)((t->n_frames + nchunks - 1) / nchunks);
        nchunks = (t->n_frames + a.frames_per_chunk - 1) / a.frames_per_chunk;
        void *d_nbuf;
        AMOF_TRY(ensure(ctx, SLOT_AUX8, per_wg * work.size() * (size_t)nchunks, &d_nbuf));
        a.nbuf = (double *)d_nbuf;
        a.ncap = (int32_t)cap;
        AMOF_TRY(clear_scratch());
        timing_dom_begin(ctx, "bad_exact_biglist");
        AMOF_HIP_TRY(ctx, launch_exact(dim3((unsigned)work.size(), (unsigned)nchunks)));
        timing_dom_end(ctx, 1);
        AMOF_TRY(read_flags(flags));
        if (flags[0]) return fail(ctx, AMOF_EANGLE, "Undefined angle");
        if (flags[1]) return fail(ctx, AMOF_EHIP, "internal error: neighbour list overflow in the big-list pass");
    }
    // complete and valid: add to the caller's (device) buffers
    if (T > 0 && t->n_frames > 0) {
        hipLaunchKernelGGL(add_u64_kernel, dim3(64), dim3(256), 0, ctx->stream, hist_dev,
                           (const unsigned long long *)d_hs, (size_t)T * KC * nb);
        hipLaunchKernelGGL(add_u64_kernel, dim3(1), dim3(64), 0, ctx->stream, nang_dev,
                           (const unsigned long long *)d_ns, (size_t)T * KC);
        AMOF_HIP_TRY(ctx, hipGetLastError());
    }
    timing_end(ctx);
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

static int bad_check(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *triples,
                     int32_t T, const double *edges, int32_t nb, const void *hist, const void *nang)
{
    AMOF_TRY(validate_traj(ctx, t, false));
    if (!cutoff || T < 0 || (T > 0 && !triples) || !edges || !hist || !nang) return fail(ctx, AMOF_EINVAL, "NULL argument");
    if (nb <= 0) return fail(ctx, AMOF_EINVAL, "nb must be positive");
    for (int k = 0; k < nb; k++)
        if (!(edges[k + 1] > edges[k])) return fail(ctx, AMOF_EINVAL, "edges must increase strictly");
    for (int k = 0; k < T; k++)
        for (int c = 0; c < 2; c++)
            if (triples[2 * k + c] < -1 || triples[2 * k + c] >= t->n_species)
                return fail(ctx, AMOF_EINVAL, "triple %d names a species out of range", k);
    return AMOF_OK;
}

extern "C" int amof_bad_hist_dev(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *triples,
                                 int32_t T, const double *edges, int32_t nb, uint64_t *hist_dev,
                                 uint64_t *n_angles_dev)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(bad_check(ctx, t, cutoff, triples, T, edges, nb, hist_dev, n_angles_dev));
    if (T == 0) return AMOF_OK;
    return bad_run(ctx, t, cutoff, triples, T, edges, nb, (unsigned long long *)hist_dev,
                   (unsigned long long *)n_angles_dev);
}

extern "C" int amof_bad_hist(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *triples,
                             int32_t T, const double *edges, int32_t nb, uint64_t *hist, uint64_t *n_angles)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(bad_check(ctx, t, cutoff, triples, T, edges, nb, hist, n_angles));
    if (T == 0) return AMOF_OK;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d_hist, *d_nang;
    size_t hb = (size_t)T * nb * sizeof(uint64_t), nbts = (size_t)T * sizeof(uint64_t);
    AMOF_TRY(upload(ctx, SLOT_OUT0, hist, hb, &d_hist));
    AMOF_TRY(upload(ctx, SLOT_OUT1, n_angles, nbts, &d_nang));
    int rc = bad_run(ctx, t, cutoff, triples, T, edges, nb, (unsigned long long *)d_hist, (unsigned long long *)d_nang);
    if (rc) return rc;
    AMOF_TRY(fetch(ctx, hist, d_hist, hb));
    AMOF_TRY(fetch(ctx, n_angles, d_nang, nbts));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

extern "C" int amof_bad_hist_by_cn(amof_ctx *ctx, const amof_traj *t, const double *cutoff, const int32_t *triples,
                                   int32_t T, const double *edges, int32_t nb, int32_t cn_max, uint64_t *hist,
                                   uint64_t *n_angles)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(bad_check(ctx, t, cutoff, triples, T, edges, nb, hist, n_angles));
    if (cn_max < 1 || cn_max > 65535) return fail(ctx, AMOF_EINVAL, "cn_max must be 1..65535");
    if (T == 0) return AMOF_OK;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d_hist, *d_nang;
    const size_t KC = (size_t)cn_max + 1;
    size_t hb = (size_t)T * KC * nb * sizeof(uint64_t), nbts = (size_t)T * KC * sizeof(uint64_t);
    AMOF_TRY(upload(ctx, SLOT_OUT0, hist, hb, &d_hist));
    AMOF_TRY(upload(ctx, SLOT_OUT1, n_angles, nbts, &d_nang));
    int rc = bad_run(ctx, t, cutoff, triples, T, edges, nb, (unsigned long long *)d_hist, (unsigned long long *)d_nang, cn_max);
    if (rc) return rc;
    AMOF_TRY(fetch(ctx, hist, d_hist, hb));
    AMOF_TRY(fetch(ctx, n_angles, d_nang, nbts));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}
