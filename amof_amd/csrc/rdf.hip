// RDF pair-distance histogram kernels (gfx950).
//
// Replaces asap3's RawRDF as driven by the reference at amof/rdf.py:88-93.  Kernel families, picked per call by
// rdf_run (amof_last_path names the one that produced the result):
//   rdf_exact      rdf_tile_kernel<ORTHO,EXTRA> / rdf_tile_kernel_global: canonical float64 arithmetic per pair and
//                  per periodic image; partially periodic cells, large image shares, nbins beyond the LDS histogram
//   rdf_tile       rdf_tile_kernel_fast<ORTHO,CULL>: 32-bit fixed-point minimum image, f32 candidate bins with an
//                  exact re-decision near every edge, slab-sorted tiles by LDS-DMA (half-cell cutoffs); general cells
//                  on the lower-triangular factor of their metric (guard_math.h)
//   rdf_tile_zf    the same for diagonal cells (the headline): ZF = f32 slab-axis coordinates per step -- the slab-sorted
//                  axis needs no per-pair wrap (with culling: every visited partner; without: all but the band around
//                  the sub-tile's antipode) -- and the always-add LDS histogram (every lane adds to its candidate bin,
//                  out-of-range pairs to trash words, flagged pairs are fixed up by the refinement: no lane masks)
//   rdf_tile_img   the same with IMG = true: cutoffs beyond half a cell height (sheared cells at the default
//                  cutoff); pairs within reach of a cell face are parked and evaluated canonically, images included
//   rdf_range      rdf_range_kernel_fast: 2-level (slab x y-bin) cell list for small cutoffs
//   rdf_cell       rdf_cell_kernel: 3-D cell list, one lane per centre atom, for cutoffs far below the cell size
//
// Common decomposition of the tile kernels: atoms are sorted by species once per call (the species of an atom never
// change along a trajectory) and cut into species-pure tiles.  A workgroup owns ONE unordered tile pair (I <= J) and
// a chunk of consecutive frames; tile J is staged in LDS and read by broadcast, and the pair's species key is uniform
// per workgroup, so a single nbins-wide u32 histogram in LDS serves the whole workgroup.  It is flushed with u64
// global atomics once per frame chunk.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "amof_internal.h"

namespace amof {

constexpr int RDF_TILE = 256;

struct RdfArgs {
    const double *pos;      // [F][N][3]
    const double *geom;     // [n_cells][GEOM_STRIDE]
    const double *img;      // [n_cells][max_img][3]
    const int32_t *nimg;    // [n_cells]
    const int32_t *perm;    // [N] species-sorted atom ids
    const Tile *tiles;
    const int2 *pairs;      // [n_pairs] (tile I, tile J), I <= J
    unsigned long long *U;  // [S*S][nbins] unordered-key histograms
    int64_t N;
    int32_t F;
    int32_t n_cells;
    int32_t frames_per_chunk;
    int32_t nbins;
    int32_t S;
    int32_t max_img;
    double rmax2;
    double dr;
};

__device__ __forceinline__ void rdf_count(unsigned *hist, double d2, double rmax2, double dr, int nbins)
{
    if (d2 < rmax2) {
        // exact IEEE sqrt and divide: the bin index is an integer result and
        // must equal the oracle's bit for bit
        int b = (int)(sqrt(d2) / dr);
        if (b < nbins) atomicAdd(&hist[b], 1u);
    }
}

template <bool ORTHO, bool EXTRA>
__global__ __launch_bounds__(RDF_TILE) void rdf_tile_kernel(RdfArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *tjx = reinterpret_cast<double *>(lds_raw);
    double *tjy = tjx + RDF_TILE;
    double *tjz = tjy + RDF_TILE;
    unsigned *hist = reinterpret_cast<unsigned *>(tjz + RDF_TILE);

    const int tid = threadIdx.x;
    const int2 pr = a.pairs[blockIdx.x];
    const Tile ti = a.tiles[pr.x];
    const Tile tj = a.tiles[pr.y];
    const bool diag = pr.x == pr.y;

    for (int k = tid; k < a.nbins; k += RDF_TILE) hist[k] = 0u;

    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, a.F);
    const int64_t ai = tid < ti.count ? a.perm[ti.start + tid] : -1;
    const int64_t aj = tid < tj.count ? a.perm[tj.start + tid] : -1;
    const double rmax2 = a.rmax2, dr = a.dr;
    const int nbins = a.nbins;

    for (int f = f0; f < f1; f++) {
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        __syncthreads();  // previous frame's tile J fully consumed (and hist zeroed)
        if (aj >= 0) {
            tjx[tid] = p[aj * 3 + 0];
            tjy[tid] = p[aj * 3 + 1];
            tjz[tid] = p[aj * 3 + 2];
        }
        double xi = 0.0, yi = 0.0, zi = 0.0;
        if (ai >= 0) {
            xi = p[ai * 3 + 0];
            yi = p[ai * 3 + 1];
            zi = p[ai * 3 + 2];
        }
        __syncthreads();
        if (ai >= 0) {
            const int jbeg = diag ? tid + 1 : 0;
            const int ne = EXTRA ? a.nimg[gi] : 0;
            const double *__restrict__ E = EXTRA ? a.img + (size_t)gi * a.max_img * 3 : nullptr;
            for (int j = jbeg; j < tj.count; j++) {
                double dx, dy, dz;
                pair_base<ORTHO>(g, tjx[j] - xi, tjy[j] - yi, tjz[j] - zi, dx, dy, dz);
                rdf_count(hist, norm2(dx, dy, dz), rmax2, dr, nbins);
                if (EXTRA) {
                    for (int m = 0; m < ne; m++)
                        rdf_count(hist, norm2(dx + E[3 * m], dy + E[3 * m + 1], dz + E[3 * m + 2]), rmax2, dr,
                                  nbins);
                }
            }
        }
    }
    __syncthreads();
    // species key (lo, hi): tiles are species-sorted so ti.species <= tj.species
    unsigned long long *U = a.U + ((size_t)ti.species * a.S + tj.species) * (size_t)nbins;
    for (int k = tid; k < nbins; k += RDF_TILE) {
        unsigned v = hist[k];
        if (v) atomicAdd(&U[k], (unsigned long long)v);
    }
}

// Same decomposition with the histogram in global memory (nbins too large for
// LDS).  Slow path, kept for completeness of the API.
template <bool ORTHO, bool EXTRA>
__global__ __launch_bounds__(RDF_TILE) void rdf_tile_kernel_global(RdfArgs a)
{
    __shared__ double tjx[RDF_TILE], tjy[RDF_TILE], tjz[RDF_TILE];
    const int tid = threadIdx.x;
    const int2 pr = a.pairs[blockIdx.x];
    const Tile ti = a.tiles[pr.x];
    const Tile tj = a.tiles[pr.y];
    const bool diag = pr.x == pr.y;
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, a.F);
    const int64_t ai = tid < ti.count ? a.perm[ti.start + tid] : -1;
    const int64_t aj = tid < tj.count ? a.perm[tj.start + tid] : -1;
    unsigned long long *U = a.U + ((size_t)ti.species * a.S + tj.species) * (size_t)a.nbins;
    for (int f = f0; f < f1; f++) {
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const int gi = a.n_cells == 1 ? 0 : f;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        __syncthreads();
        if (aj >= 0) {
            tjx[tid] = p[aj * 3 + 0];
            tjy[tid] = p[aj * 3 + 1];
            tjz[tid] = p[aj * 3 + 2];
        }
        double xi = 0.0, yi = 0.0, zi = 0.0;
        if (ai >= 0) {
            xi = p[ai * 3 + 0];
            yi = p[ai * 3 + 1];
            zi = p[ai * 3 + 2];
        }
        __syncthreads();
        if (ai >= 0) {
            const int jbeg = diag ? tid + 1 : 0;
            const int ne = EXTRA ? a.nimg[gi] : 0;
            const double *__restrict__ E = EXTRA ? a.img + (size_t)gi * a.max_img * 3 : nullptr;
            for (int j = jbeg; j < tj.count; j++) {
                double dx, dy, dz;
                pair_base<ORTHO>(g, tjx[j] - xi, tjy[j] - yi, tjz[j] - zi, dx, dy, dz);
                double d2 = norm2(dx, dy, dz);
                if (d2 < a.rmax2) {
                    int b = (int)(sqrt(d2) / a.dr);
                    if (b < a.nbins) atomicAdd(&U[b], 1ull);
                }
                if (EXTRA) {
                    for (int m = 0; m < ne; m++) {
                        double e2 = norm2(dx + E[3 * m], dy + E[3 * m + 1], dz + E[3 * m + 2]);
                        if (e2 < a.rmax2) {
                            int b = (int)(sqrt(e2) / a.dr);
                            if (b < a.nbins) atomicAdd(&U[b], 1ull);
                        }
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------
// Fast path (fully periodic cells whose cutoff needs no extra images).
//
// quantize_kernel folds every atom into the cell once per frame and stores its
// fractional coordinates as 32-bit fixed point, species-sorted (16 B / atom).
// In the pair loop the minimum image is then free -- the u32 difference wraps --
// and an f32 candidate bin q~ = sqrt(d2~)/dr is accepted only when it is
// provably the canonical one:  |q~ - q| <= g_f  with
//     g_f = nbins * eps_f + g_m,   g_m = quant / dr + nbins * 1e-12,   quant = 2^-31 * (|c0|+|c1|+|c2|)
// eps_f = fast_guard_rel() bounds the relative error of the f32 chain (cvt, scale, squares/fma and
// v_sqrt_f32; 3.3e-7 for diagonal cells, (5 kappa + 3.06) * 2^-24 * 1.1 for sheared ones); the
// fixed-point grid moves a distance by < quant.  A lane whose q~ lies within g_f of an integer (or
// of nbins) is refined: first with the f64 distance of the same integer differences (decides unless
// within g_m of the edge), then with the canonical f64 arithmetic and exact sqrt/divide.  At
// |s_k| = 1/2 the wrapped image and the canonical one may differ: in a diagonal cell both have the
// same length; sheared cells take the fast path only when the cutoff stays clear of every half
// height, so both images are out of range there.
// per-cell record of the fast path (one per frame when the cell changes)
struct FrameScale {
    float sc[9];        // ORTHO: sc[0..2] = L_k * 2^-32 / dr, sc[3..5] their squares ; else the lower-triangular factor of the
                        // metric of cell * 2^-32 / dr (rows in stored order): sc[0], sc[3], sc[4], sc[6..8]; the others 0
    uint32_t cull_gap;  // slab-gap threshold of this cell (0 = culling off)
    uint32_t near_t[3]; // IMG variant: a pair with |i_k| > near_t[k] (stored axis order) is evaluated canonically
    uint32_t _pad;
    double sc64[9];     // the same factors in f64 (level-2 refinement)
    // TRI variant (see RdfTri below): sc[3..5] = L00^2, L11^2, L22^2, sc[8] = L22 (units: bins per 2^-32 of the axis)
    uint32_t tri_kx, tri_ky;    // round(kx 2^32), round(ky 2^32) mod 2^32: what one cell of slab wrap adds to the folded x'', y'
    float tri_c10;              // L10 / L00
    float tri_near_y;           // a pair with |fl(iy)| > tri_near_y may have a second image in range (+inf: never)
    float tri_near_z;           // likewise |dz| (bins) > tri_near_z
    float tri_near_x;           // likewise |fl(ix + c10 iy)| (slow path only: the variant is refused when the fast path would need it)
};

struct RdfFastArgs {
    RdfArgs a;
    const QAtom *Q;          // [nf][N] species-sorted, slab-sorted
    const FrameScale *fs;    // [n_cells]
    int32_t f_base;          // first frame of this batch
    int32_t nf;              // frames in this batch
    float half_m_guard;      // 1/2 - g_f rounded DOWN (g_f in bins: the f32 candidate's error bound)
    float nb_hi;             // nbins + g_f rounded UP
    double guard64;          // g_m (bins): f64-from-fixed-point candidate
    double guard64_2;        // 2 g_m (1 + 1e-9), g_m^2 (1 + 1e-9): level 2 decides on T - e^2 against +-(2 e g_m + g_m^2)
    double guard64_sq;
    int32_t xcd_map;         // 1: chunk -> XCD affinity mapping of the grid
    int32_t n_chunks;        // tile kernel: frames [c nf_main / n_chunks, (c+1) nf_main / n_chunks) belong to chunk c
    int32_t nf_main;         // tile kernel: frames dealt in chunks (nf, or with the XCD mapping the multiple of 8 below it: the
                             // nf % 8 frames behind them are rows n_chunks .. of the grid, one frame each, on whatever XCD)
    int32_t img_queue;       // IMG variant: capacity of one parking buffer
    int32_t img_defer;       // 1: two small buffers, a step's parked pairs are evaluated at the start of the next step
                             // (no extra barrier); 0: one large buffer, evaluated at the end of the step
};

constexpr int FAST_THREADS = 256;
constexpr int IMG_QUEUE_MAX = 1024; // IMG variant: parked near-face pairs per step and buffer (capacity chosen by the host)
constexpr int FAST_TILE = 512;      // two centre atoms per thread
constexpr int FAST_TRASH = 32;      // tile kernel: words behind the LDS histogram that take the out-of-range pairs (fast_bin<AA>)

// a wave-uniform double, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// candidate bin coordinate q~ = |r_j - r_i| / dr from the fixed-point fractional coordinates
template <bool ORTHO>
__device__ __forceinline__ float fast_q(const float *sc, int ix, int iy, int iz)
{
    const float fx = (float)ix, fy = (float)iy, fz = (float)iz;
    float t;
    if (ORTHO) {
        // squares first, then the squared scales sc[3..5]: 7u on t instead of 9u (see fast_guard_rel)
        const float x2 = fx * fx, y2 = fy * fy, z2 = fz * fz;
        t = fmaf(z2, sc[5], fmaf(y2, sc[4], x2 * sc[3]));
    } else {
        // general cell: the scales are the lower-triangular factor of the cell's metric (lower_factor, amof_internal.h)
        const float dx = fmaf(fz, sc[6], fmaf(fy, sc[3], fx * sc[0]));
        const float dy = fmaf(fz, sc[7], fy * sc[4]);
        const float dz = fz * sc[8];
        t = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
    return __builtin_amdgcn_sqrtf(t);
}

// ZF form (diagonal cell, slab-culled tile pairs): the slab-axis difference needs no per-pair wrap -- the partners a
// step visits lie within reach of the centre sub-tile's slab range -- so it arrives as the difference dz of two f32
// values already in units of bins (partner and centre relative to the sub-tile's slab midpoint, one rounding each):
// v_sub_f32 + v_fmac_f32 instead of v_sub_u32 + v_cvt_f32_i32 + v_mul_f32 + v_fmac_f32.  Error bound: fast_guard_zf.
__device__ __forceinline__ float fast_q_zf(const float *sc, int ix, int iy, float dz)
{
    const float fx = (float)ix, fy = (float)iy;
    const float x2 = fx * fx, y2 = fy * fy;
    return __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(y2, sc[4], x2 * sc[3])));
}

// squared candidate coordinate (bins^2) in f64 from the fixed-point differences
template <bool ORTHO>
__device__ __forceinline__ double medium_t(const double *sc, int ix, int iy, int iz)
{
    const double fx = (double)ix, fy = (double)iy, fz = (double)iz;
    double dx, dy, dz;
    if (ORTHO) {
        dx = fx * sc[0]; dy = fy * sc[1]; dz = fz * sc[2];
    } else {
        dx = fma(fz, sc[6], fma(fy, sc[3], fx * sc[0]));
        dy = fma(fz, sc[7], fy * sc[4]);
        dz = fz * sc[8];
    }
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

// Refinement of a pair whose f32 candidate q sits within g_f of the bin edge
// e = rint(q).  Level 2 decides on which side of e the pair lies from the f64
// squared distance T of the same fixed-point differences (error: the 2^-32 grid
// only): T >= (e+g_m)^2 -> bin e, T < (e-g_m)^2 -> bin e-1 (bin nbins = out of
// range).  Level 3 (|q - e| <= g_m, or coincident atoms): canonical arithmetic
// on the original float64 positions with exact sqrt and divide.
// (ZF kernels keep the partner's f32 slab coordinate in qj.w: the atom index then comes from the quantised frame, qseg[j])
// PROV: the fast path has already counted the pair in its candidate bin (int)q (always-add scheme): take that back.
template <bool ORTHO, bool ZF = false, bool PROV = false>
__device__ __forceinline__ void rdf_pair_refine(unsigned *hist, const RdfFastArgs &fa,
                                                const double *sc64, const double *__restrict__ g,
                                                float q, uint32_t uix, uint32_t uiy, uint32_t uiz, uint4 qj,
                                                const double *__restrict__ p, uint32_t idx_i,
                                                const QAtom *__restrict__ qseg = nullptr, int j = 0)
{
    const int ix = (int)(qj.x - uix), iy = (int)(qj.y - uiy), iz = (int)(qj.z - uiz);
    const double T = medium_t<ORTHO>(sc64, ix, iy, iz);
    const float ef = rintf(q);
    const double e = (double)ef;
    // T >= (e + g_m)^2  <=>  D = T - e^2 >= 2 e g_m + g_m^2 =: band (D is exact: e^2 is an integer below 2^24 and T
    // - e^2 fits the significand); T < (e - g_m)^2 is implied by D < -band (stricter by 2 g_m^2: a few more level-3 pairs)
    const double D = fma(-e, e, T), band = fma(e, fa.guard64_2, fa.guard64_sq);
    const int ei = (int)ef;
    int b = -1;
    if (D >= band) b = ei;
    else if (D < -band && ei > 0) b = ei - 1;
    if (PROV) {
        const int cand = (int)q;                       // (q < nbins + 1/2 here: never clamped)
        if (b == cand) return;                         // the provisional count was right
        atomicAdd(&hist[cand], 0xffffffffu);           // (u32 counters wrap; the sum is what is flushed)
    }
    if (b >= 0) {
        if (b < fa.a.nbins) atomicAdd(&hist[b], 1u);
        return;
    }
    const uint32_t idx_j = ZF ? qseg[j].idx : qj.w;
    const double *pi = p + (size_t)idx_i * 3, *pj = p + (size_t)idx_j * 3;
    double dx, dy, dz;
    pair_base<ORTHO>(g, pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2], dx, dy, dz);
    rdf_count(hist, norm2(dx, dy, dz), fa.a.rmax2, fa.a.dr, fa.a.nbins);
}

// Canonical evaluation of a whole pair -- the base image and every listed further image, exactly as the exact
// kernels (and the oracle) do it.  Used by the IMG variant for the pairs that lie within reach of a cell face.
template <bool ORTHO>
__device__ __forceinline__ void rdf_pair_images(unsigned *hist, const RdfFastArgs &fa, const double *__restrict__ g,
                                                const double *__restrict__ p, uint32_t idx_i, uint32_t idx_j, int gi)
{
    const double *pi = p + (size_t)idx_i * 3, *pj = p + (size_t)idx_j * 3;
    double dx, dy, dz;
    pair_base<ORTHO>(g, pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2], dx, dy, dz);
    rdf_count(hist, norm2(dx, dy, dz), fa.a.rmax2, fa.a.dr, fa.a.nbins);
    const int ne = fa.a.max_img > 0 ? fa.a.nimg[gi] : 0;
    const double *__restrict__ E = fa.a.img + (size_t)gi * fa.a.max_img * 3;
    for (int m = 0; m < ne; m++)
        rdf_count(hist, norm2(dx + E[3 * m], dy + E[3 * m + 1], dz + E[3 * m + 2]), fa.a.rmax2, fa.a.dr, fa.a.nbins);
}

// One pair on the fast path.  With g = guard_f: a candidate whose fractional
// part is > g away from both bin edges is certainly in bin (int)q -- and if
// that bin is >= nbins the pair is certainly out of range; everything else
// below nbins + g is refined.  Returns true when the pair needs refinement.
// IMG: the cutoff reaches beyond half a cell height (or too close to it): pairs whose fractional difference lies
// within reach of a cell face (|i_k| > near_t[k]) can have a second image in range or an ambiguous base image --
// they are not touched here (near = true) and are evaluated canonically, images included; for every other pair the
// base image is unambiguous and the only one that can be in range.
// IMG variant: does the pair lie within reach of a cell face, |i_k| > near_t[k] on some axis?  Decided on the f32
// conversions with thresholds near_f[k] just below float(near_t[k]) (rounded down, then one ulp less): a pair beyond the
// integer threshold is always flagged, a few just below it may be too -- they take the canonical route, which is exact.
__device__ __forceinline__ bool near_pair(const float *near_f, int ix, int iy, int iz)
{
    return (fabsf((float)ix) > near_f[0]) | (fabsf((float)iy) > near_f[1]) | (fabsf((float)iz) > near_f[2]);
}

// The always-add scheme's clamp (see fast_bin<AA>): out-of-range and dead candidates go to the trash words -- and, for
// bin_count's conversion, candidates below a quarter of a bin are raised to 1/4 (such a pair is in bin 0 whatever a
// refinement would say: its true q is < 1/4 + guard).  One v_med3_f32 where a v_min_f32 stood.
__device__ __forceinline__ float clamp_candidate(float q, float clampv)
{
    return __builtin_amdgcn_fmed3f(q, 0.25f, clampv);
}

// hist[floor(q)] += 1 for a clamped candidate, 1/4 <= q < 2^20, without the half-rate conversion: 4 q - 1/2 + 2^23 (one fma,
// one rounding; the sum lies in [2^23, 2^24), where floats are the integers) rounds to floor(4 q) -- a tie exists only where
// 4 q is an integer, and goes to the even neighbour: 4 q itself or 4 q - 1, which lie above the same multiple of four unless q
// is an integer, and there the even neighbour IS 4 q -- so the low bits of the sum are floor(4 q) and (bits & 0x7ffffc) is
// the counter's byte offset 4 floor(q).  v_fmaak_f32 + v_and_b32 (2.2 issue cycles each) instead of v_cvt_i32_f32 +
// v_lshl_add_u32 (4.1 each): the headline launch 72.0 -> 69.9 ms with the base still added, the first move of this kernel since round 2
// (profiles/r05/tile_bin_address.txt; round 3 had tried 4 q through the conversion and an integer mask: slower).
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ void bin_count(unsigned *hist, float q)
{
    // (the histogram is the first thing in the kernel's dynamic LDS and the kernel has no static LDS: its byte offset IS its
    //  LDS address -- checked on the host before every launch, allow_max_lds_from_zero -- so no base is added)
    (void)hist;
    const unsigned off = __float_as_uint(fmaf(q, 4.0f, 8388607.5f)) & 0x007ffffcu;
    lds_u32 *p = (lds_u32 *)(uintptr_t)off;
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool ORTHO, bool IMG = false, bool ZF = false, bool AA = false>
__device__ __forceinline__ bool fast_bin(unsigned *hist, const float *sc, bool live, float half_m_guard,
                                         float nb_hi, uint32_t uix, uint32_t uiy, uint32_t uiz, uint4 qj,
                                         float &q, const float *near_f = nullptr, bool *near = nullptr,
                                         float zif = 0.0f, float clampv = 0.0f, bool live_all = false)
{
    const int ix = (int)(qj.x - uix), iy = (int)(qj.y - uiy), iz = (int)(qj.z - uiz);
    if (IMG) {
        // on the converted differences the candidate needs anyway: one compare per axis (thresholds rounded down, see
        // the kernel; axes that are clear of their half height carry +inf)
        const bool nr = near_pair(near_f, ix, iy, iz);
        *near = live && nr;
        live = live && !nr;
    }
    q = ZF ? fast_q_zf(sc, ix, iy, __uint_as_float(qj.w) - zif) : fast_q<ORTHO>(sc, ix, iy, iz);
    if (AA) {
        // Always add, fix up later (tile kernel): one LDS atomic costs the CU's LDS pipe the same whatever the number
        // of active lanes (profiles/r02/ubench_lds_atomic.txt), so nothing is gained by masking -- and the masks (two
        // compares, five scalar instructions and a skip branch per pair) were what the loop waited for.  Every lane
        // adds to the candidate bin of min(q, clampv): clampv = nbins + 1/2 + (lane mod 32) sends out-of-range (and
        // dead: q = inf) pairs to 32 trash words behind the histogram with a "safe" fractional part, so that `unsafe`
        // is exactly "in range and within the guard of a bin edge"; such a pair has been counted provisionally and
        // rdf_pair_refine<PROV> takes that count back before it adds the exact one.
        if (!live_all) q = live ? q : __builtin_inff();
        q = clamp_candidate(q, clampv);
        const bool unsafe = !(fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < half_m_guard);
        bin_count(hist, q);
        return unsafe;
    }
    const bool in = live && (q < nb_hi);
    const bool safe = fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < half_m_guard;
    if (in && safe) atomicAdd(&hist[(int)q], 1u);
    return in && !safe;
}

// a lane's two-deep queue of pairs that wait for their refinement (PARK; see fast_quad)
struct LaneQueue {
    unsigned pk0, pk1;      // (partner index in tile J) << 1 | (centre b?); QUEUE_EMPTY: free
    float pq0, pq1;         // the pair's clamped f32 candidate
};
constexpr unsigned QUEUE_EMPTY = 0xffffffffu;

template <bool ORTHO, bool DIAG, bool TAIL, bool IMG = false, bool ZF = false, bool ZFK = false, bool AA = false, bool PARK = false>
__device__ __forceinline__ void fast_quad(unsigned *hist, const RdfFastArgs &fa, const double *sc64,
                                          const double *__restrict__ g, const float *sc, const uint4 *tq,
                                          int j0, int cntj, bool has_a, bool has_b, int ia, int ib,
                                          float half_m_guard, float nb_hi, uint32_t uax, uint32_t uay,
                                          uint32_t uaz, uint32_t ida, uint32_t ubx, uint32_t uby, uint32_t ubz,
                                          uint32_t idb, const double *__restrict__ p,
                                          const float *near_f = nullptr, int gi = 0, uint2 *nq = nullptr,
                                          unsigned *nq_count = nullptr, unsigned nq_cap = 0,
                                          float zaf = 0.0f, float zbf = 0.0f, const QAtom *__restrict__ qseg = nullptr,
                                          float clampv = 0.0f, LaneQueue *lq = nullptr)
{
    // ZF: this step uses the f32 slab coordinates; ZFK: the kernel is a ZF kernel (the .w of the LDS copies may have
    // been overwritten by an earlier step of the frame: atom indices always come from the quantised frame)
    // four partner atoms per trip, read by broadcast before any LDS atomic
    uint4 qj[4];
#pragma unroll
    for (int u = 0; u < 4; u++) qj[u] = tq[j0 + u];
    float qa[4], qb[4];
    bool na[4], nb[4];   // per-pair "needs refinement" flags (kept as lane masks)
    bool anynear = false;   // IMG: some pair of this quad lies within reach of a cell face (one mask only: SGPRs are scarce)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = j0 + u;
        const bool la = has_a && (!TAIL || j < cntj) && (!DIAG || j > ia);
        const bool lb = has_b && (!TAIL || j < cntj) && (!DIAG || j > ib);
        bool ma = false, mb = false;
        // (ZF: centres that do not exist carry an infinite slab coordinate, so only DIAG / TAIL need a live mask)
        na[u] = fast_bin<ORTHO, IMG, ZF, AA>(hist, sc, la, half_m_guard, nb_hi, uax, uay, uaz, qj[u], qa[u], near_f, &ma, zaf,
                                             clampv, ZF && !DIAG && !TAIL);
        nb[u] = fast_bin<ORTHO, IMG, ZF, AA>(hist, sc, lb, half_m_guard, nb_hi, ubx, uby, ubz, qj[u], qb[u], near_f, &mb, zbf,
                                             clampv, ZF && !DIAG && !TAIL);
        anynear |= ma | mb;
    }
    if (na[0] | na[1] | na[2] | na[3] | nb[0] | nb[1] | nb[2] | nb[3]) {   // a few % of the pairs
        if (PARK) {
            // PARK: a flagged pair waits in its lane's two-deep register queue for the end of the step, where ONE body per
            // queue level serves every lane that has an entry (half the lanes at the first level) -- instead of one body per
            // pair slot of every quad that has a flagged lane (~0.5 bodies of ~100 issue cycles per quad, each for one or two
            // live lanes).  A lane whose queue is full keeps the old way for that pair (ovf).
            unsigned ovf = 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (na[u]) {
                    if (lq->pk1 != QUEUE_EMPTY) ovf |= 1u << (2 * u);
                    else { lq->pk1 = lq->pk0; lq->pq1 = lq->pq0; lq->pk0 = (unsigned)(j0 + u) << 1; lq->pq0 = qa[u]; }
                }
                if (nb[u]) {
                    if (lq->pk1 != QUEUE_EMPTY) ovf |= 2u << (2 * u);
                    else { lq->pk1 = lq->pk0; lq->pq1 = lq->pq0; lq->pk0 = ((unsigned)(j0 + u) << 1) | 1u; lq->pq0 = qb[u]; }
                }
            }
            if (ovf) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ovf & (1u << (2 * u))) rdf_pair_refine<ORTHO, ZFK, AA>(hist, fa, sc64, g, qa[u], uax, uay, uaz, qj[u], p, ida, qseg, j0 + u);
                    if (ovf & (2u << (2 * u))) rdf_pair_refine<ORTHO, ZFK, AA>(hist, fa, sc64, g, qb[u], ubx, uby, ubz, qj[u], p, idb, qseg, j0 + u);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (na[u]) rdf_pair_refine<ORTHO, ZFK, AA>(hist, fa, sc64, g, qa[u], uax, uay, uaz, qj[u], p, ida, qseg, j0 + u);
                if (nb[u]) rdf_pair_refine<ORTHO, ZFK, AA>(hist, fa, sc64, g, qb[u], ubx, uby, ubz, qj[u], p, idb, qseg, j0 + u);
            }
        }
    }
    if (IMG) {
        if (anynear) {   // rare: find the pairs again (cheap) and park them for the dense canonical pass of this step
#pragma unroll 1
            for (int u = 0; u < 4; u++) {
                const uint4 q = tq[j0 + u];      // (from LDS again: indexing the register copies would spill them)
                const int j = j0 + u;
                const bool la = has_a && (!TAIL || j < cntj) && (!DIAG || j > ia);
                const bool lb = has_b && (!TAIL || j < cntj) && (!DIAG || j > ib);
                auto is_near = [&](uint32_t ux, uint32_t uy, uint32_t uz) {
                    return near_pair(near_f, (int)(q.x - ux), (int)(q.y - uy), (int)(q.z - uz));
                };
                auto park = [&](uint32_t idc) {
                    const unsigned slot = atomicAdd(nq_count, 1u);
                    if (slot < nq_cap) nq[slot] = make_uint2(idc, q.w);
                    else rdf_pair_images<ORTHO>(hist, fa, g, p, idc, q.w, gi);   // queue full: evaluate in place
                };
                if (la && is_near(uax, uay, uaz)) park(ida);
                if (lb && is_near(ubx, uby, ubz)) park(idb);
            }
        }
    }
}

// --------------------------------------------------------------------------
// TRI: general (triclinic) cells in the orthogonalised lattice frame.
//
// With the cell vectors in stored order A, B, C (C the slab axis) and L the lower-triangular factor of their metric,
// an image of a pair has the components  Z = L22 fz,  Y = L11 (fy + r21 fz),  X = L00 (fx + c10 fy + r20 fz)
// (c10 = L10/L00, r20 = L20/L00, r21 = L21/L11).  quantize_kernel stores FOLDED coordinates x'' = x + kx z, y' = y + ky z
// (kx = r20 - c10 r21, ky = r21), so the wrapped u32 differences of a pair are  iy = fy + r21 fz  and
// ix = fx + r20 fz - c10 r21 fz, i.e.  Y = L11 iy,  X = L00 (ix + c10 iy)  -- one fma more than a diagonal cell -- of the
// image whose |Y| and |ix| are smallest (the slab difference is the minimum image by the culling window, or wrapped).
// The fold uses every atom's own z in [0, 1): a pair whose slab difference wraps m cells (m = -1, 0, 1) is corrected
// by m (Kx, Ky) -- per piece of the partner window on the centre's side where the slab difference comes from f32
// coordinates (ZF quads), per pair from the sign and the borrow of the slab subtraction otherwise (tri_int).
// Which pairs are decided here: an image with |d| < R has |Z|, |Y|, |X| < R.  The host admits the variant when
//   L22 >= 2R or near_z is flagged,  L11 >= 2R or near_y is flagged,  R / L00 + |c10| / 2 < 1/2
// (R = rmax with the guards): then the image above is the ONLY one that can be in range unless |Y| > L11 - R or
// |Z| > L22 - R ("near": flagged by one compare each, taken back and evaluated canonically with every listed image),
// and the x wrap -- decided on ix without the c10 iy term -- can only go wrong where both candidates are out of range.
// Levels: f32 candidate (always-add), f64 T of the same integers, canonical evaluation (parked, drained densely).
__device__ __forceinline__ void tri_int(const uint4 qj, uint32_t ux, uint32_t uy, uint32_t uz, uint32_t kx, uint32_t ky,
                                        int &ix, int &iy, int &iz)
{
    iz = (int)(qj.z - uz);
    // unwrapped slab difference = wrapped + m: m = (wrapped < 0) - (borrow)
    const uint32_t sgn = (uint32_t)(iz >> 31);           // all ones when the wrapped difference is negative
    const bool borrow = qj.z < uz;
    ix = (int)(qj.x - ux - (sgn & kx) + (borrow ? kx : 0u));
    iy = (int)(qj.y - uy - (sgn & ky) + (borrow ? ky : 0u));
}

// XW: the x wrap is decided WITH the c10 iy term (cells whose x axis has no slack, e.g. two equal in-plane lengths):
// the term joins the integer difference before it wraps (truncated product: the candidate moves by < 1 + 256 |c10| grid
// units, part of the guards; level 2 repeats the decision bit for bit and is exact for the image it picks)
template <bool XW>
__device__ __forceinline__ int tri_xwrap(float c10, int ix, float fy)
{
    return XW ? (int)((uint32_t)ix + (uint32_t)(int)(fy * c10)) : ix;
}

// squared distance of a candidate in bins^2 (f32) | its root
template <bool XW>
__device__ __forceinline__ float tri_t(const float *sc, float c10, int ix, float fy, float dz)
{
    const float dxf = XW ? (float)tri_xwrap<true>(c10, ix, fy) : fmaf(fy, c10, (float)ix);
    const float x2 = dxf * dxf, y2 = fy * fy;
    return fmaf(dz, dz, fmaf(y2, sc[4], x2 * sc[3]));
}

template <bool XW>
__device__ __forceinline__ float tri_q(const float *sc, float c10, int ix, float fy, float dz)
{
    return __builtin_amdgcn_sqrtf(tri_t<XW>(sc, c10, ix, fy, dz));
}

// near mode 4: the nearer of the pair's two candidates -- the image it minimised and the one a cell further along y, x wrapped
// again with the new y.  The host admits the mode only where the two differ by a lattice vector of length >= 2 rmax: then at
// most one of them is in range, and when the nearer one is clear of the last bin edge by the guard the other is out of range
// by the same margin (a pair inside that band is flagged and goes the canonical way with every listed image, like any pair on
// a bin edge).  One root, one clamp, one edge test, one histogram add per pair; the first version counted both candidates
// (the second into the trash words for the 85 % of pairs that have none): 2.3x the diagonal cell's cost per visited pair.
// HALF = +-1: c10 = +-1/2 exactly (hexagonal cells: two equal in-plane vectors at 120 or 60 degrees).  The y term of the x
// wrap is then iy / 2 -- an arithmetic shift and an integer add where the float product, its conversion and the add stood --
// and the twin image's x, a further half cell along, is the first candidate's with the top bit flipped (+-2^31 modulo 2^32);
// the twin's |y| is | |iy| - 2^32 |.  Exact to one grid unit (the float product: 1 + 256 |c10| units, which the guard keeps).
template <bool XW, int HALF>
__device__ __forceinline__ float tri_q_twin(const float *sc, float c10, int ix, int iy, float fy, float dz)
{
    if (HALF != 0) {
        const uint32_t h = (uint32_t)(iy >> 1);
        const uint32_t x1 = HALF > 0 ? (uint32_t)ix + h : (uint32_t)ix - h;
        const float f1 = (float)(int)x1, f2 = (float)(int)(x1 ^ 0x80000000u);
        const float y2 = fabsf(fy) - 4294967296.f;
        const float t1 = fmaf(dz, dz, fmaf(fy * fy, sc[4], (f1 * f1) * sc[3]));
        const float t2 = fmaf(dz, dz, fmaf(y2 * y2, sc[4], (f2 * f2) * sc[3]));
        return __builtin_amdgcn_sqrtf(__builtin_fminf(t1, t2));
    }
    const float t1 = tri_t<XW>(sc, c10, ix, fy, dz);
    const float t2 = tri_t<true>(sc, c10, ix, fy - copysignf(4294967296.f, fy), dz);
    return __builtin_amdgcn_sqrtf(__builtin_fminf(t1, t2));
}

struct TriConst {
    float c10, near_y, near_z, near_x;
    uint32_t kx, ky;
};

// one pair on the TRI fast path (always-add histogram, see fast_bin<AA>); returns "needs the slow path".
// ZF: (ux, uy) are the centre's folded coordinates already corrected for the piece's slab wrap, zif its f32 slab
// coordinate; else the raw folded ones (per-pair correction).  NEAR: 0 no second image possible anywhere; 1 only for pairs
// inside the guard band of the last bin edge, which are flagged anyway: the slow path tests, the fast path does not; 2 the
// fast path tests y, 3 y and z (the slow path always both: every instruction of its body costs, so NEAR = 0 has none);
// 4 for cells whose second image along y is COMMON (hexagonal: 15 % of the pairs): the fast path takes the nearer of the
// pair's two candidates (tri_q_twin), and a pair whose nearer candidate is inside the guard of a bin edge is parked for the
// canonical arithmetic (first version: the twin in the slow path, which then ran on every wave-level pair with a few live
// lanes: 4.4x the diagonal cell's cost per visited pair; second: both candidates counted, 2.3x).
template <bool ZF, int NEAR, bool XW, int HALF = 0>
__device__ __forceinline__ bool fast_bin_tri(unsigned *hist, const float *sc, const TriConst &tc, bool live, float half_m_guard,
                                             uint32_t ux, uint32_t uy, uint32_t uz, uint4 qj, float &q, float zif,
                                             float clampv, bool live_all)
{
    int ix, iy;
    float dz;
    if (ZF) {
        ix = (int)(qj.x - ux); iy = (int)(qj.y - uy);
        dz = __uint_as_float(qj.w) - zif;
    } else {
        int iz;
        tri_int(qj, ux, uy, uz, tc.kx, tc.ky, ix, iy, iz);
        dz = (float)iz * sc[8];
    }
    const float fy = (float)iy;
    static_assert(HALF == 0 || NEAR == 4, "the exact-half x wrap exists for near mode 4 only");
    q = NEAR == 4 ? tri_q_twin<XW, HALF>(sc, tc.c10, ix, iy, fy, dz) : tri_q<XW>(sc, tc.c10, ix, fy, dz);
    if (!live_all) q = live ? q : __builtin_inff();
    q = clamp_candidate(q, clampv);
    bool flag = !(fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < half_m_guard);
    if (NEAR == 2) flag |= live && (fabsf(fy) > tc.near_y);
    if (NEAR == 3) flag |= live && ((fabsf(fy) > tc.near_y) | (fabsf(dz) > tc.near_z));
    bin_count(hist, q);
    return flag;
}

// slow path of a flagged TRI pair: the provisional count of candidate bin (int)q stands, moves, or is taken back and the
// pair parked for the canonical evaluation (near pairs: a second image may count; level 3: within g_m of a bin edge)
// (ZF quads: (ux, uy) carry the piece's slab wrap -- for a partner inside the window that is the pair's own; one outside it is
//  out of range by its slab distance alone, whatever the other two components come out as)
template <bool ZF, int NEAR, bool XW, typename Park>
__device__ __forceinline__ void rdf_pair_refine_tri(unsigned *hist, const RdfFastArgs &fa, const float *sc, const TriConst &tc,
                                                    const double *sc64, float q, uint32_t ux, uint32_t uy, uint32_t uz,
                                                    uint4 qj, float zif, float clampv, Park &&park)
{
    int ix, iy, iz;
    if (ZF) { ix = (int)(qj.x - ux); iy = (int)(qj.y - uy); iz = (int)(qj.z - uz); }
    else tri_int(qj, ux, uy, uz, tc.kx, tc.ky, ix, iy, iz);
    const int cand = (int)q;
    if (NEAR == 4) {
        // the nearer candidate is within the guard of a bin edge: its provisional count back, the pair goes the canonical way
        // with every listed image
        atomicAdd(&hist[cand], 0xffffffffu);
        park();
        return;
    }
    // (+inf on an axis without second image; x: only pairs inside the guard band of the last bin edge can matter)
    if (NEAR > 0 && ((fabsf((float)iy) > tc.near_y) | (fabsf((float)iz * sc[8]) > tc.near_z) |
                     (fabsf(XW ? (float)tri_xwrap<true>(tc.c10, ix, (float)iy) : fmaf((float)iy, tc.c10, (float)ix)) > tc.near_x))) {
        atomicAdd(&hist[cand], 0xffffffffu);
        park();
        return;
    }
    double fx = (double)ix;
    const double fy = (double)iy, fz = (double)iz;
    if (XW) {   // the image the fast path picked: the same f32 product decides the wrap, the shift is a whole cell
        const double it = (double)(int)((float)iy * tc.c10);
        fx += (double)tri_xwrap<true>(tc.c10, ix, (float)iy) - (fx + it);      // (-2^32, 0 or 2^32: exact)
    }
    const double dx = fma(fy, sc64[3], fx * sc64[0]), dy = fy * sc64[4], dz = fz * sc64[8];
    const double T = fma(dz, dz, fma(dy, dy, dx * dx));
    const float ef = rintf(q);
    const double e = (double)ef;
    const double D = fma(-e, e, T), band = fma(e, fa.guard64_2, fa.guard64_sq);
    const int ei = (int)ef;
    int b = -1;
    if (D >= band) b = ei;
    else if (D < -band && ei > 0) b = ei - 1;
    if (b == cand) return;
    atomicAdd(&hist[cand], 0xffffffffu);
    if (b >= 0) {
        if (b < fa.a.nbins) atomicAdd(&hist[b], 1u);
        return;
    }
    park();
}

// the atom index of partner j of the tile (wave-uniform address): a SCALAR load, so that the slow path holds no vector-memory
// operation that the compiler could make the quad loops wait for
__device__ __forceinline__ uint32_t scalar_idx(const QAtom *q)
{
    const unsigned long long a = (unsigned long long)q;
    const unsigned long long au = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0xc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(au) : "memory");
    return v;
}

template <bool DIAG, bool TAIL, bool ZF, int NEAR, bool XW, int HALF = 0>
__device__ __forceinline__ void fast_quad_tri(unsigned *hist, const RdfFastArgs &fa, const double *sc64, const float *sc,
                                              const TriConst &tc, const uint4 *tq, int j0, int cntj, bool has_a, bool has_b,
                                              int ia, int ib, float half_m_guard,
                                              uint32_t uax, uint32_t uay, uint32_t uaz, uint32_t ida,
                                              uint32_t ubx, uint32_t uby, uint32_t ubz, uint32_t idb,
                                              uint2 *nq, unsigned *nq_count, unsigned nq_cap,
                                              const double *__restrict__ g, const double *__restrict__ p, int gi,
                                              float zaf, float zbf, const QAtom *__restrict__ qseg, float clampv)
{
    uint4 qj[4];
#pragma unroll
    for (int u = 0; u < 4; u++) qj[u] = tq[j0 + u];
    float qa[4], qb[4];
    bool na[4], nb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = j0 + u;
        const bool la = has_a && (!TAIL || j < cntj) && (!DIAG || j > ia);
        const bool lb = has_b && (!TAIL || j < cntj) && (!DIAG || j > ib);
        na[u] = fast_bin_tri<ZF, NEAR, XW, HALF>(hist, sc, tc, la, half_m_guard, uax, uay, uaz, qj[u], qa[u], zaf, clampv, ZF && !DIAG && !TAIL);
        nb[u] = fast_bin_tri<ZF, NEAR, XW, HALF>(hist, sc, tc, lb, half_m_guard, ubx, uby, ubz, qj[u], qb[u], zbf, clampv, ZF && !DIAG && !TAIL);
    }
    if (na[0] | na[1] | na[2] | na[3] | nb[0] | nb[1] | nb[2] | nb[3]) {
        unsigned ovf = 0u;      // pairs of this lane that found the queue full: bit 2 u + (0: centre a, 1: centre b)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            auto park = [&](uint32_t idc) {
                const uint32_t idj = scalar_idx(qseg + (j0 + u));
                const unsigned slot = atomicAdd(nq_count, 1u);
                if (slot < nq_cap) nq[slot] = make_uint2(idc, idj);
                else ovf |= 1u << (2 * u + (idc == idb ? 1 : 0));       // (evaluated below: ONE copy of the canonical body per quad loop, not eight)
            };
            // (the partner's record is read from LDS again here: four records held in registers across the eight slow-path
            //  bodies of a quad are 16 of the 96 VGPRs, and the TRI variants spilled 136 - 240 bytes per lane into these loops)
            if (na[u] | nb[u]) {
                const uint4 qr = tq[j0 + u];
                if (na[u]) rdf_pair_refine_tri<ZF, NEAR, XW>(hist, fa, sc, tc, sc64, qa[u], uax, uay, uaz, qr, zaf, clampv, [&]() { park(ida); });
                if (nb[u]) rdf_pair_refine_tri<ZF, NEAR, XW>(hist, fa, sc, tc, sc64, qb[u], ubx, uby, ubz, qr, zbf, clampv, [&]() { park(idb); });
            }
        }
        if (ovf) {      // queue full (perfect lattices: every pair on a bin edge): in place
#pragma unroll 1
            for (int k = 0; k < 8; k++)
                if ((ovf >> k) & 1u) rdf_pair_images<false>(hist, fa, g, p, (k & 1) ? idb : ida, scalar_idx(qseg + (j0 + (k >> 1))), gi);
        }
    }
}

// Work item = (tile I, 128-atom sub-tile of I, tile J) x a chunk of frames.  All four
// waves of the workgroup hold the SAME 128 centre atoms (two adjacent ones per lane) and
// share the quads of tile J round-robin, so the slab culling -- which depends only on the
// centre atoms' slab range -- removes the same share of work from every wave.
constexpr int FAST_SUB = 128;

// One wave copies 64 x 16 B from per-lane global addresses to 1 KiB of LDS at `dst`
// (wave-uniform), without passing through registers (LDS-DMA, global_load_lds_dwordx4).
__device__ __forceinline__ void dma_1k(const QAtom *src_lane, uint4 *dst_wave)
{
    dma16(src_lane, dst_wave);
}

template <bool ORTHO, bool CULL, bool IMG = false, bool ZFK = false, int TRI = -1>
// (five workgroups per CU, 96 VGPRs.  The TRI variants once spilled 136 - 240 bytes per lane at that budget -- 7x the vector-
//  memory instructions of the diagonal kernel -- and the ones with near tests in the fast path ran better at four workgroups
//  and 128 VGPRs; since their slow path reads the partner's record from LDS again and the canonical fallback of a full queue
//  exists once per quad loop instead of eight times, all of them fit: profiles/r04/tri_experiments.txt)
__global__ __launch_bounds__(FAST_THREADS, 5) void rdf_tile_kernel_fast(RdfFastArgs fa)
{
    // TRI >= 0: general cells in the orthogonalised lattice frame (fast_quad_tri; TRI % 5: near tests, TRI / 5: x wrap with the y term)
    static_assert(!ZFK || (ORTHO && !IMG) || TRI >= 0, "f32 slab coordinates: diagonal cells (no image queue) or TRI");
    static_assert(TRI < 0 || (!ORTHO && !IMG && ZFK), "TRI: general cells, f32 slab coordinates, its own queue");
    // always-add histogram scheme (fast_bin<AA>): measured per variant (profiles/r02/tile_variants.txt) -- the plain
    // general-cell variant spills under it (nine scales, 96 VGPRs) and keeps the masked form
    constexpr bool AA = ORTHO || IMG;
    constexpr bool QUEUE = IMG || TRI >= 0;     // parked pairs, drained densely by the canonical arithmetic
    // flagged pairs wait in per-lane register queues for the end of the step (fast_quad): the diagonal-cell kernels.  68.6 ->
    // 67.0 ms per headline launch -- the first parking scheme that pays (seven with LDS queues lost: profiles/r04/rdf_tile_stop.txt)
    constexpr bool PARK = ORTHO && ZFK && !IMG && TRI < 0 && AA;
    // TRI = 10 / 11: near mode 4 with the exact-half x wrap, c10 = + 1/2 / - 1/2 (tri_q_twin)
    constexpr int NEAR = TRI >= 10 ? 4 : (TRI >= 0 ? TRI % 5 : 0);
    constexpr bool XW = TRI >= 5;
    constexpr int HALF = TRI == 10 ? 1 : (TRI == 11 ? -1 : 0);
    const RdfArgs &a = fa.a;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // double-buffered tiles: J (512 entries) and the centre sub-tile (128 entries)
    // (the histogram first: bin_count forms LDS byte addresses from the candidate's bits)
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);                  // [nbins + trash], padded to 16 bytes
    uint4 *tqb = reinterpret_cast<uint4 *>(hist + ((fa.a.nbins + FAST_TRASH + 2 + 3) & ~3));      // [2][FAST_TILE]
    uint4 *tcb = tqb + 2 * FAST_TILE;                                        // [2][FAST_SUB]
    // IMG: queue of the pairs that need the canonical evaluation (drained densely once per step), two counters
    // (two buffers: a step parks into one while the previous step's is drained; three counters so that one can be
    // reset a whole step away from its last reader and its next writer)
    uint2 *nq_base = reinterpret_cast<uint2 *>(tcb + 2 * FAST_SUB);                             // [2][img_queue]
    const unsigned nq_cap = (unsigned)fa.img_queue;
    // (the three counters behind the queue, in the dynamic LDS like everything else: the kernel has NO static LDS, so that
    //  the dynamic part starts at LDS address 0 -- bin_count)
    unsigned *nq_count = reinterpret_cast<unsigned *>(nq_base + (size_t)(fa.img_defer ? 2 : 1) * nq_cap);        // [3]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the quad loops below count in SGPRs)
    // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (id % 8), each
    // with its own 4 MiB L2.  All work items of one frame chunk go to the same XCD so that the
    // chunk's quantised frames (16 frames x 16 B x N) are re-read from that L2, not from the
    // fabric.  Pure bijection of the grid: correctness does not depend on the placement.
    // The frames that do not divide by 8 (nf % 8 of them) come last, one frame per grid row, their work items on all XCDs:
    // kept in the XCDs' ranges they made one XCD's share a whole frame longer (625 frames: 79 against 78 1/8, 1.1 % of a
    // rank's launch in an 8-GPU run).
    unsigned chunk = blockIdx.y, bx = blockIdx.x;
    const bool tail = (int)blockIdx.y >= fa.n_chunks;
    if (fa.xcd_map && !tail) {      // (off when there are too few chunks to give every XCD its share)
        const unsigned long long lin = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = (unsigned)(lin & 7ull);
        const unsigned long long kk = lin >> 3;
        chunk = (unsigned)(kk / gridDim.x) * 8u + xcd;
        bx = (unsigned)(kk % gridDim.x);
    }
    const int2 pr = a.pairs[bx];
    const Tile ti = a.tiles[pr.x];
    const Tile tj = a.tiles[pr.y];
    const bool diag = pr.x == pr.y;
    const int nbins = a.nbins;
    for (int k = tid; k < nbins; k += FAST_THREADS) hist[k] = 0u;

    // equal shares (+-1 frame): with the XCD mapping n_chunks is a multiple of 8, so every XCD gets the same work
    // (with the XCD mapping, XCD x owns the contiguous frame range [x nf/8, (x+1) nf/8), cut into n_chunks/8 shares)
    const unsigned cs = fa.xcd_map ? (chunk & 7u) * ((unsigned)fa.n_chunks >> 3) + (chunk >> 3) : chunk;
    const int f0 = tail ? fa.nf_main + ((int)blockIdx.y - fa.n_chunks) : (int)((long long)cs * fa.nf_main / fa.n_chunks);
    const int f1 = tail ? f0 + 1 : (int)((long long)(cs + 1) * fa.nf_main / fa.n_chunks);
    // The (up to four) 128-atom centre sub-tiles of tile I are handled by the same workgroup, frame by
    // frame: a step is (frame, sub-tile); tile J of a frame is staged once and serves all its sub-tiles,
    // and the LDS histogram is flushed once for everything.
    const int nsub = (ti.count + FAST_SUB - 1) / FAST_SUB;
    const int la = 2 * lane, lb = la + 1;                          // local indices in the sub-tile
    const float half_m_guard = fa.half_m_guard;
    const float nb_hi = fa.nb_hi;
    const float clampv = (float)nbins + 0.5f + (float)(lane & (FAST_TRASH - 1));   // see fast_bin<AA>
    const int cntj = tj.count;
    const int cntj4 = (cntj + 3) & ~3;
    const int full = cntj & ~3;

    // LDS-DMA (1 KiB per wave instruction), indices clamped (slots beyond the counts are never used unmasked):
    // tile J of frame fl into J buffer jb -- wave w moves entries [64w, 64w+64) and [256+64w, ...)
    auto stage_j = [&](int fl, int jb) {
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        uint4 *tq = tqb + jb * FAST_TILE;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int k = r * 256 + wave * 64 + lane;
            if (r * 256 + wave * 64 < cntj4) dma_1k(Qf + tj.start + min(k, cntj - 1), tq + r * 256 + wave * 64);
        }
    };
    // centre sub-tile `sub` of frame fl into centre buffer cb -- waves 0/1
    auto stage_c = [&](int fl, int sub, int cb) {
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        const int cnt = min(FAST_SUB, ti.count - sub * FAST_SUB);
        if (wave < 2 && wave * 64 < cnt)
            dma_1k(Qf + ti.start + sub * FAST_SUB + min(wave * 64 + lane, cnt - 1), tcb + cb * FAST_SUB + wave * 64);
    };

    const int nsteps = (f1 - f0) * nsub;
    if (QUEUE && tid < 3) nq_count[tid] = 0u;
    const double *p_prev = nullptr, *g_prev = nullptr;
    int gi_prev = 0;
    if (nsteps > 0) { stage_j(f0, 0); stage_c(f0, 0, 0); }
    float sc[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    double sc64r[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    TriConst trc = {0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
    float near_f[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    uint32_t cull_gap = 0u;
    for (int step = 0, fl = f0, sub = 0; step < nsteps; step++) {
        const int jb = (fl - f0) & 1, cbuf = step & 1;
        const int cnti = min(FAST_SUB, ti.count - sub * FAST_SUB);     // centre atoms of this sub-tile
        const int ia = sub * FAST_SUB + la, ib = ia + 1;               // indices in tile I
        const bool has_a = la < cnti, has_b = lb < cnti;
        const int f = fa.f_base + fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const uint4 *tq = tqb + jb * FAST_TILE, *tc = tcb + cbuf * FAST_SUB;
        const int gi = a.n_cells == 1 ? 0 : f;
        const FrameScale *__restrict__ fs = fa.fs + gi;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        // wave-uniform per-cell constants: kept in scalar registers; a constant cell's are read once, at the first step
        // (per step they cost ~20 v_readfirstlane and a global-memory latency at the head of every step)
        if (step == 0 || a.n_cells != 1) {
            cull_gap = __builtin_amdgcn_readfirstlane(fs->cull_gap);
#pragma unroll
            for (int k = 0; k < 9; k++)
                sc[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->sc[k])));
            // the f64 scales of the level-2 refinement: diagonal cells keep their three in scalar registers (a vector load
            // at the head of every refinement visit costs a memory latency each time)
            if (ORTHO) {
#pragma unroll
                for (int k = 0; k < 3; k++) sc64r[k] = uniform_f64(fs->sc64[k]);
            }
            if (TRI >= 0) {     // L00, L10, L11, L22 of the level-2 form (rdf_pair_refine_tri): a load at the head of every visit would cost its latency
                sc64r[0] = uniform_f64(fs->sc64[0]); sc64r[3] = uniform_f64(fs->sc64[3]);
                sc64r[4] = uniform_f64(fs->sc64[4]); sc64r[8] = uniform_f64(fs->sc64[8]);
                trc.c10 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->tri_c10)));
                trc.near_y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->tri_near_y)));
                trc.near_z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->tri_near_z)));
                trc.near_x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->tri_near_x)));
                trc.kx = __builtin_amdgcn_readfirstlane(fs->tri_kx);
                trc.ky = __builtin_amdgcn_readfirstlane(fs->tri_ky);
            }
            if (IMG) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t tk = __builtin_amdgcn_readfirstlane(fs->near_t[k]);
                    near_f[k] = tk != 0x7fffffffu ? __uint_as_float(__float_as_uint(__uint2float_rd(tk)) - 1u) : __builtin_inff();
                }
            }
        }
        const double *sc64 = (ORTHO || TRI >= 0) ? sc64r : fs->sc64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for this step has landed
        __syncthreads();                                    // everyone's has; the previous step is fully consumed
        if (QUEUE) {
            // what the previous step parked is complete (barrier above): dense canonical pass, one pair per lane,
            // while this step parks into the other buffer; the counter of step + 1 was last read a step ago
            if (tid == 0) nq_count[(step + 1) % 3] = 0u;
            if (fa.img_defer && step > 0) {
                const uint2 *nqp = nq_base + (size_t)((step - 1) & 1) * nq_cap;
                const int npark = (int)min(nq_count[(step - 1) % 3], nq_cap);
                for (int e = tid; e < npark; e += FAST_THREADS) {
                    const uint2 pr2 = nqp[e];
                    rdf_pair_images<ORTHO>(hist, fa, g_prev, p_prev, pr2.x, pr2.y, gi_prev);
                }
            }
            p_prev = p; g_prev = g; gi_prev = gi;
        }
        uint2 *nq = nq_base + (fa.img_defer ? (size_t)(step & 1) * nq_cap : 0);
        // what the next step needs streams in behind the arithmetic
        {
            const int nsub_next = sub + 1 < nsub ? sub + 1 : 0;
            const int fl_next = sub + 1 < nsub ? fl : fl + 1;
            if (step + 1 < nsteps) stage_c(fl_next, nsub_next, cbuf ^ 1);
            if (sub == 0 && fl + 1 < f1) stage_j(fl + 1, jb ^ 1);   // (buffer jb^1: frame fl-1, consumed before this barrier)
        }
        const uint4 ca = tc[min(la, cnti - 1)], cb = tc[min(lb, cnti - 1)];
        const uint32_t uax = ca.x, uay = ca.y, uaz = ca.z, ida = ca.w;
        const uint32_t ubx = cb.x, uby = cb.y, ubz = cb.z, idb = cb.w;
        // Partner index range(s) to visit.  Tile J is slab-sorted along the stored .z axis, so
        // the partners within reach of the centre atoms (slab distance <= cull_gap, circular)
        // form at most two contiguous index ranges, found once per frame.
        int rb0 = diag ? (sub * FAST_SUB) : 0, re0 = cntj, rb1 = 0, re1 = 0;
        bool zf = false;          // this step has quads that run on f32 slab coordinates (ZF kernels)
        uint32_t z0 = 0u;         // their origin: the middle of the centre sub-tile's slab range
        int mz[2] = {0, 0};       // TRI: cells of slab wrap between the centres and the ZF partners of piece 0 / 1
        bool touch = false;       // TRI: the two pieces of a wrapped reach touch; [sa, sb) = the quads that straddle them
        int sa = 0, sb = 0;
        // quad ranges [b, e) of this step, quads dealt round-robin to the four waves: zq on f32 slab coordinates,
        // iq on the integer slab differences
        int zqb[2] = {0, 0}, zqe[2] = {0, 0}, iqb[2] = {0, 0}, iqe[2] = {0, 0};
        if (CULL || ZFK) {
            // the sub-tile is slab-sorted: its first / last atoms give its slab range
            const uint32_t s_first = tc[0].z >> 24, s_last = tc[cnti - 1].z >> 24;
            const uint32_t wlo = s_first << 24, whi = (s_last << 24) | 0xffffffu;
            const uint32_t W = whi - wlo;
            // every lane samples two quads of tile J; ballots give the boundary quads
            const int q0 = lane, q1 = lane + 64;     // FAST_TILE / 4 = 128 quads
            const uint32_t l0 = tq[min(4 * q0 + 3, cntj - 1)].z >> 24, l1 = tq[min(4 * q1 + 3, cntj - 1)].z >> 24;
            const uint32_t h0 = tq[min(4 * q0, cntj - 1)].z >> 24, h1 = tq[min(4 * q1, cntj - 1)].z >> 24;
            auto first_set = [&](bool g0, bool g1) {
                const unsigned long long m0 = __ballot(g0), m1 = __ballot(g1);
                const int fq = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : 128);
                return min(4 * fq, cntj);
            };
            if (CULL) {
                const uint32_t G = cull_gap;
                // reachable keys: [wlo - G, whi + G] (mod 2^32), widened to whole slabs (2^24 each)
                const unsigned long long span = (unsigned long long)W + 2ull * G + (2ull << 24);
                if (G != 0u && span < (1ull << 32)) {
                    const uint32_t klo = wlo - G, khi = whi + G;
                    const uint32_t slo = klo >> 24, shi = khi >> 24;
                    // a_: start of the first quad whose LAST partner has slab >= slo
                    // b_: start of the first quad whose FIRST partner has slab > shi
                    const int a_ = first_set(4 * q0 >= cntj || l0 >= slo, 4 * q1 >= cntj || l1 >= slo);
                    const int b_ = first_set(4 * q0 >= cntj || h0 > shi, 4 * q1 >= cntj || h1 > shi);
                    if (klo <= khi) {
                        rb0 = max(rb0, a_); re0 = b_;
                    } else if (b_ < a_) {      // wrapped reach: keys <= khi or >= klo
                        re0 = b_;
                        rb1 = max(rb0, a_); re1 = cntj;
                        // (piece 0 = the keys <= khi: a cell below the centres when whi + G ran over; piece 1 = the keys
                        //  >= klo: a cell above them when wlo - G ran under)
                        if (khi < whi) mz[0] = -1;
                        if (wlo < G) mz[1] = 1;
                    } else if (TRI >= 0) {
                        // the two pieces touch (no partner in the gap, or the tile lies inside one of them).  TRI needs the
                        // slab wrap of a piece: quads before a_ hold keys <= khi only (or the gap), quads from b_ on keys >=
                        // klo only; the quads in between (b_ >= a_) straddle both and take the integer path
                        touch = true;
                        re0 = a_;
                        rb1 = max(rb0, b_); re1 = cntj;
                        sa = max(rb0, a_); sb = b_;
                        if (khi < whi) mz[0] = -1;
                        if (wlo < G) mz[1] = 1;
                    }                          // else the two pieces touch: whole tile
                    if (ZFK) {
                        // f32 slab coordinates are valid when no slab difference of a partner inside the reach window and
                        // a centre can wrap: |z_j - z0| <= G + W/2 + 2^24 + 1, |z_i - z0| <= W/2 + 1, their difference
                        // below 2^31 in magnitude -- then it IS the minimum image.  A partner outside the window (quads are
                        // visited whole) comes out beyond G either way round, i.e. out of range.  W < 2^28 is the span
                        // the host's error bound (fast_guard_zf) assumes.
                        zf = W < (1u << 28) && (unsigned long long)G + W + (2ull << 24) < (1ull << 31);
                        z0 = wlo + (W >> 1);
                    }
                }
                int qbr[2], qer[2];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int rb = r == 0 ? rb0 : rb1, re = r == 0 ? re0 : re1;
                    qbr[r] = rb & ~3;
                    qer[r] = re <= rb ? 0 : (re + 3) & ~3;
                    if (r == 1) qbr[r] = max(qbr[r], (max(re0, rb0) + 3) & ~3);   // never visit a quad twice
                    (zf ? zqb : iqb)[r] = qbr[r];
                    (zf ? zqe : iqe)[r] = qer[r];
                }
                if (TRI >= 0 && touch) {
                    if (zf) {               // (the integer ranges are free in a ZF step)
                        iqb[0] = sa & ~3; iqe[0] = sb <= sa ? 0 : (sb + 3) & ~3;
                    } else {                // everything on the integer path: one range
                        iqb[0] = rb0 & ~3; iqe[0] = (cntj + 3) & ~3;
                        iqb[1] = 0; iqe[1] = 0;
                    }
                }
            } else {
                // No culling (the cutoff reaches across the slab axis: cubic cells at the default cutoff): every quad is
                // visited, and the f32 slab coordinates still hold for all partners but a band around the antipode of
                // the sub-tile, where the wrap depends on the centre.  G2 = the largest reach they allow; a ZF quad may
                // only hold partners inside the window (here a partner outside is NOT out of range), so the window is
                // cut at whole quads from the inside: a2 = first quad whose FIRST partner has slab >= slo, b2 = first
                // quad whose LAST partner has slab > shi.  The band (>= 4 slabs wide) and the straddling quads take the
                // integer path.
                const int start = rb0, endq = (cntj + 3) & ~3;          // (start is a multiple of 4)
                iqb[0] = start; iqe[0] = endq;
                if (W < (1u << 28)) {
                    const uint32_t G2 = 0x7fffffffu - W - (2u << 24) - 2u;
                    const uint32_t klo = wlo - G2, khi = whi + G2;
                    const uint32_t slo = klo >> 24, shi = khi >> 24;
                    const int a2 = first_set(4 * q0 >= cntj || h0 >= slo, 4 * q1 >= cntj || h1 >= slo);
                    const int b2 = first_set(4 * q0 >= cntj || l0 > shi, 4 * q1 >= cntj || l1 > shi);
                    auto up4 = [](int x) { return (x + 3) & ~3; };
                    zf = true;
                    z0 = wlo + (W >> 1);
                    if (klo <= khi) {              // window = slabs [slo, shi]: ZF quads [a2, b2)
                        zqb[0] = max(start, a2); zqe[0] = up4(b2);
                        iqb[0] = start; iqe[0] = min(max(start, a2), endq);
                        iqb[1] = max(start, up4(b2)); iqe[1] = endq;
                    } else {                       // window = slabs [0, shi] and [slo, 255]: ZF quads [0, b2) and [a2, end)
                        zqb[0] = start; zqe[0] = up4(b2);
                        zqb[1] = max(start, a2); zqe[1] = endq;
                        iqb[0] = max(start, up4(b2)); iqe[0] = min(max(start, a2), endq);
                        if (khi < whi) mz[0] = -1;
                        if (wlo < G2) mz[1] = 1;
                    }
                }
            }
        } else {
            iqb[0] = rb0 & ~3;
            iqe[0] = (re0 + 3) & ~3;
        }
        float zaf = 0.0f, zbf = 0.0f;
        if (ZFK && zf) {
            // each wave converts the slab coordinate of the partners it is about to meet on the ZF path (its own quads
            // of both pieces) into bins relative to z0, one rounding (f64 product -> f32), and parks it in the LDS
            // copy's .w; centre atoms likewise, in registers.  Wave-local: LDS operations of a wave execute in order.
            const double cz = TRI >= 0 ? sc64[8] : sc64[2];
            uint4 *tqw = tqb + jb * FAST_TILE;
#pragma unroll 1
            for (int r = 0; r < 2; r++) {
                for (int j = zqb[r] + 4 * wave + 16 * (lane >> 2) + (lane & 3); j < zqe[r]; j += 256)
                    tqw[j].w = __float_as_uint((float)((double)(int)(tqw[j].z - z0) * cz));
            }
            zaf = has_a ? (float)((double)(int)(uaz - z0) * cz) : __builtin_inff();
            zbf = has_b ? (float)((double)(int)(ubz - z0) * cz) : __builtin_inff();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const QAtom *__restrict__ qseg = fa.Q + (size_t)fl * (size_t)a.N + tj.start;
        LaneQueue lq = {QUEUE_EMPTY, QUEUE_EMPTY, 0.0f, 0.0f};
        auto run = [&](auto zf_tag, const int *qbr, const int *qer) {
            constexpr bool ZF = decltype(zf_tag)::value;
#pragma unroll 1
            for (int r = 0; r < 2; r++) {
                const int qb = qbr[r], qe = qer[r];
                if (qe <= qb) continue;
                const int qe_full = min(qe, full);
                if (TRI >= 0) {
                    // ZF quads: the piece's slab wrap goes onto the centres' folded coordinates; integer quads correct per pair
                    const uint32_t kxm = ZF ? (uint32_t)mz[r] * trc.kx : 0u, kym = ZF ? (uint32_t)mz[r] * trc.ky : 0u;
                    const uint32_t cax = uax + kxm, cay = uay + kym, cbx = ubx + kxm, cby = uby + kym;
                    unsigned *nqc = &nq_count[step % 3];
                    if (diag) {
                        for (int j0 = qb + 4 * wave; j0 < qe; j0 += 16)
                            fast_quad_tri<true, true, ZF, NEAR, XW, HALF>(hist, fa, sc64, sc, trc, tq, j0, cntj, has_a, has_b, ia, ib, half_m_guard,
                                                                cax, cay, uaz, ida, cbx, cby, ubz, idb,
                                                                nq, nqc, nq_cap, g, p, gi, zaf, zbf, qseg, clampv);
                    } else {
                        int j0 = qb + 4 * wave;
                        for (; j0 < qe_full; j0 += 16)
                            fast_quad_tri<false, false, ZF, NEAR, XW, HALF>(hist, fa, sc64, sc, trc, tq, j0, cntj, has_a, has_b, ia, ib, half_m_guard,
                                                                  cax, cay, uaz, ida, cbx, cby, ubz, idb,
                                                                  nq, nqc, nq_cap, g, p, gi, zaf, zbf, qseg, clampv);
                        if (j0 == full && j0 < qe && full < cntj)
                            fast_quad_tri<false, true, ZF, NEAR, XW, HALF>(hist, fa, sc64, sc, trc, tq, full, cntj, has_a, has_b, ia, ib, half_m_guard,
                                                                 cax, cay, uaz, ida, cbx, cby, ubz, idb,
                                                                 nq, nqc, nq_cap, g, p, gi, zaf, zbf, qseg, clampv);
                    }
                    continue;
                }
                if (diag) {
                    for (int j0 = qb + 4 * wave; j0 < qe; j0 += 16)
                        fast_quad<ORTHO, true, true, IMG, ZF, ZFK, AA, PARK>(hist, fa, sc64, g, sc, tq, j0, cntj, has_a, has_b, ia, ib,
                                                                   half_m_guard, nb_hi, uax, uay, uaz, ida, ubx, uby, ubz,
                                                                   idb, p, near_f, gi, nq, &nq_count[step % 3], nq_cap, zaf,
                                                                   zbf, qseg, clampv, &lq);
                } else {
                    int j0 = qb + 4 * wave;
                    for (; j0 < qe_full; j0 += 16)
                        fast_quad<ORTHO, false, false, IMG, ZF, ZFK, AA, PARK>(hist, fa, sc64, g, sc, tq, j0, cntj, has_a, has_b, ia, ib,
                                                                     half_m_guard, nb_hi, uax, uay, uaz, ida, ubx, uby, ubz,
                                                                     idb, p, near_f, gi, nq, &nq_count[step % 3], nq_cap,
                                                                     zaf, zbf, qseg, clampv, &lq);
                    if (j0 == full && j0 < qe && full < cntj)
                        fast_quad<ORTHO, false, true, IMG, ZF, ZFK, AA, PARK>(hist, fa, sc64, g, sc, tq, full, cntj, has_a, has_b, ia, ib,
                                                                    half_m_guard, nb_hi, uax, uay, uaz, ida, ubx, uby, ubz,
                                                                    idb, p, near_f, gi, nq, &nq_count[step % 3], nq_cap, zaf,
                                                                    zbf, qseg, clampv, &lq);
                }
            }
        };
        if (ZFK && zf) run(std::true_type{}, zqb, zqe);
        run(std::false_type{}, iqb, iqe);
        if (PARK) {
            // the step's parked pairs: one refinement body per queue level, for every lane that has an entry there (the
            // partner's record from the tile's LDS copy again, the centre by the entry's low bit)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const unsigned code = k == 0 ? lq.pk0 : lq.pk1;
                const float qk = k == 0 ? lq.pq0 : lq.pq1;
                if (code != QUEUE_EMPTY) {
                    const int j = (int)(code >> 1);
                    const bool isb = (code & 1u) != 0u;
                    const uint4 qr = tq[j];
                    rdf_pair_refine<ORTHO, ZFK, AA>(hist, fa, sc64, g, qk, isb ? ubx : uax, isb ? uby : uay, isb ? ubz : uaz, qr, p,
                                                    isb ? idb : ida, qseg, j);
                }
            }
        }
        if (QUEUE && !fa.img_defer) {
            // large shares: dense canonical pass over this step's parked pairs right away (one more barrier per step)
            __syncthreads();
            const int npark = (int)min(nq_count[step % 3], nq_cap);
            for (int e = tid; e < npark; e += FAST_THREADS) {
                const uint2 pr2 = nq[e];
                rdf_pair_images<ORTHO>(hist, fa, g, p, pr2.x, pr2.y, gi);
            }
        }
        if (++sub == nsub) { sub = 0; fl++; }
    }
    if (QUEUE && fa.img_defer && nsteps > 0) {   // the last step's parked pairs
        __syncthreads();
        const uint2 *nqp = nq_base + (size_t)((nsteps - 1) & 1) * nq_cap;
        const int npark = (int)min(nq_count[(nsteps - 1) % 3], nq_cap);
        for (int e = tid; e < npark; e += FAST_THREADS) {
            const uint2 pr2 = nqp[e];
            rdf_pair_images<ORTHO>(hist, fa, g_prev, p_prev, pr2.x, pr2.y, gi_prev);
        }
    }
    __syncthreads();
    unsigned long long *U = a.U + ((size_t)ti.species * a.S + tj.species) * (size_t)nbins;
    for (int k = tid; k < nbins; k += FAST_THREADS) {
        unsigned v = hist[k];
        if (v) atomicAdd(&U[k], (unsigned long long)v);
    }
}

// --------------------------------------------------------------------------
// Range kernel: small cutoffs in big cells (rmax <= a third of the slab axis).
//
// Frames come 2-level sorted (quantize2_kernel): nz slabs (>= rmax thick) x 256 y-bins.  One
// workgroup = one 128-atom centre sub-tile x one partner species; instead of meeting fixed
// partner tiles it walks the partner ranges itself: for each slab within one slab of the
// centres, the y-bins within rmax of the centres' y range give at most two contiguous index
// ranges (table start2), streamed through LDS 512 at a time.  Same-species pairs are counted
// once with a per-lane rule that needs no look-back: partner slab == centre slab + 1 (mod nz):
// every pair; same slab: partner index > centre index; anything else is the other atom's job
// (or farther than a whole slab: out of range).  Needs nz >= 3.
struct RdfRangeArgs {
    RdfFastArgs f;
    const uint32_t *start2;   // [nf][S][nz*256 + 1]
    const int64_t *sp_first;  // [S+1]
    const int4 *work;         // (centre species a, first centre c0, partner species b, 0)
    int32_t nz;
    int32_t gy_bins;          // y reach in bins (rmax / h_y * 256, rounded up, + 1)
};

template <bool ORTHO>
__global__ __launch_bounds__(FAST_THREADS, 5) void rdf_range_kernel_fast(RdfRangeArgs ra)
{
    const RdfFastArgs &fa = ra.f;
    const RdfArgs &a = fa.a;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    uint4 *tq = reinterpret_cast<uint4 *>(lds_raw);                  // [FAST_TILE] partner chunk
    unsigned *hist = reinterpret_cast<unsigned *>(tq + FAST_TILE);    // [nbins]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int4 w = ra.work[blockIdx.x];
    const int sa_ = w.x, c0 = w.y, sb_ = w.z;
    const bool same = sa_ == sb_;
    const int nz = ra.nz, nkeys = nz * 256;
    const int nbins = a.nbins;
    for (int k = tid; k < nbins; k += FAST_THREADS) hist[k] = 0u;
    const int64_t segA = ra.sp_first[sa_], segB = ra.sp_first[sb_];
    const int nA = (int)(ra.sp_first[sa_ + 1] - segA);
    const int cnt_c = min(FAST_SUB, nA - c0);
    const int la = 2 * lane, lb = la + 1;
    const bool has_a = la < cnt_c, has_b = lb < cnt_c;
    const float half_m_guard = fa.half_m_guard;
    const float nb_hi = fa.nb_hi;
    const int f0 = blockIdx.y * a.frames_per_chunk;
    const int f1 = min(f0 + a.frames_per_chunk, fa.nf);

    for (int fl = f0; fl < f1; fl++) {
        const int f = fa.f_base + fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        const int gi = a.n_cells == 1 ? 0 : f;
        const FrameScale *__restrict__ fs = fa.fs + gi;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        float sc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) sc[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->sc[k])));
        const uint32_t *__restrict__ stB = ra.start2 + ((size_t)fl * a.S + sb_) * (size_t)(nkeys + 1);
        // this lane's two centre atoms and their slabs
        const QAtom ca = Qf[segA + c0 + min(la, cnt_c - 1)], cb = Qf[segA + c0 + min(lb, cnt_c - 1)];
        const int za = (int)__umulhi(ca.uz, (unsigned)nz), zb = (int)__umulhi(cb.uz, (unsigned)nz);
        // slab / y-bin extent of the sub-tile (it is sorted by (slab, ybin): first and last atom)
        const QAtom cf = Qf[segA + c0], cl = Qf[segA + c0 + cnt_c - 1];
        const int s_first = (int)__umulhi(cf.uz, (unsigned)nz), s_last = (int)__umulhi(cl.uz, (unsigned)nz);
        const bool one_slab = s_first == s_last;
        int ylo = 0, yhi = 255;
        bool y_full = true;
        if (one_slab) {
            const int y0 = (int)(cf.uy >> 24) - ra.gy_bins, y1 = (int)(cl.uy >> 24) + ra.gy_bins;
            if (y1 - y0 + 1 < 256) { y_full = false; ylo = y0 & 255; yhi = y1 & 255; }
        }
        // partner slabs: [s_first - 1 (different species only), s_last + 1], each at most once
        int p_begin = same ? s_first : s_first - 1;
        int p_count = s_last + 1 - p_begin + 1;
        if (p_count > nz) { p_count = nz; }
        for (int pi = 0; pi < p_count; pi++) {
            const int ps = ((p_begin + pi) % nz + nz) % nz;
            // per-lane rule for this partner slab: threshold on the partner's sorted index
            //   -1: every partner counts, INT_MAX: none, i: partners with index > i
            auto thresh = [&](int zc, int iabs) {
                const int d = ((ps - zc) % nz + nz) % nz;
                if (same) return d == 1 ? -1 : (d == 0 ? iabs : 0x7fffffff);
                return (d <= 1 || d == nz - 1) ? -1 : 0x7fffffff;
            };
            const int tha = thresh(za, c0 + la), thb = thresh(zb, c0 + lb);
            // y ranges of this slab: [ylo, yhi] circular -> one or two index ranges
            int rb[2], re[2];
            const uint32_t *row = stB + ps * 256;
            if (y_full || ylo <= yhi) {
                rb[0] = (int)row[y_full ? 0 : ylo]; re[0] = (int)row[(y_full ? 255 : yhi) + 1];
                rb[1] = re[1] = 0;
            } else {
                rb[0] = (int)row[0]; re[0] = (int)row[yhi + 1];
                rb[1] = (int)row[ylo]; re[1] = (int)row[256];
            }
            for (int r = 0; r < 2; r++) {
                for (int base = rb[r]; base < re[r]; base += FAST_TILE) {
                    const int cntj = min(FAST_TILE, re[r] - base);
                    const int cntj4 = (cntj + 3) & ~3;
                    __syncthreads();
                    for (int k = tid; k < cntj4; k += FAST_THREADS) {
                        const QAtom q = Qf[segB + base + min(k, cntj - 1)];
                        tq[k] = make_uint4(q.ux, q.uy, q.uz, q.idx);
                    }
                    __syncthreads();
                    // per-lane thresholds relative to this chunk (fast_quad's "j > ia" form)
                    const int ia = tha == -1 ? -1 : (tha == 0x7fffffff ? 0x7fffffff : tha - base);
                    const int ib = thb == -1 ? -1 : (thb == 0x7fffffff ? 0x7fffffff : thb - base);
                    for (int j0 = 4 * wave; j0 < cntj4; j0 += 16)
                        fast_quad<ORTHO, true, true>(hist, fa, fs->sc64, g, sc, tq, j0, cntj, has_a, has_b, ia, ib, half_m_guard,
                                                     nb_hi, ca.ux, ca.uy, ca.uz, ca.idx, cb.ux, cb.uy, cb.uz, cb.idx, p);
                }
            }
        }
    }
    __syncthreads();
    const int lo = min(sa_, sb_), hi = max(sa_, sb_);
    unsigned long long *U = a.U + ((size_t)lo * a.S + hi) * (size_t)nbins;
    for (int k = tid; k < nbins; k += FAST_THREADS) {
        unsigned v = hist[k];
        if (v) atomicAdd(&U[k], (unsigned long long)v);
    }
}

// --------------------------------------------------------------------------
// Cell-list kernel: cutoffs far below the cell size (rmax <= h_k / 2.5 on every axis).
//
// Frames come sorted into a 3-D grid of cells at least rmax/2 thick (launch_cell_sort), x
// fastest, species-sorted inside a cell.  One lane = one centre atom; it walks the 13 rows
// (dz, dy) of the positive half shell -- (0,0), (0,1), (0,2), (1,-2..2), (2,-2..2) -- and in
// each row the x-contiguous run of cells cx-2 .. cx+2 (row (0,0): from its own successor to
// the end of cell cx+2), i.e. at most two index ranges per row (periodic wrap in x).  Every
// unordered pair of atoms within reach is met exactly once (the grid has >= 5 cells per axis, so
// +d and -d never name the same cell).  Partners are gathered from L2 (a frame's records fit in
// it and all workgroups of a frame are dealt to one XCD); the pair arithmetic is the fast
// path's three levels; the LDS histogram holds every unordered species pair.
struct RdfCellArgs {
    RdfFastArgs f;
    const uint32_t *start3;   // [nf][nkeys + 1]
    const uint32_t *keyoff;   // [S*S] offset of the pair (sa, sb) in the LDS histogram (pair key * nbins)
    const uint32_t *keyU;     // [npk] offset of pair key k in U ((lo * S + hi) * nbins)
    int32_t nx, ny, nz, npk;
    int32_t frames_grid;      // frame slots of the launch (rounded up to a multiple of 8 with the XCD mapping)
    int32_t fpc;              // frames per workgroup (round 4): the same block of atoms over fpc frames, ONE histogram flush
    int32_t cpt;              // centre atoms per thread (the workgroup covers 256 * cpt consecutive atoms)
    int32_t trim;             // 1: rows trimmed per lane to the cells within reach (AMOF_RDF_NOTRIM=1: the full 5-cell runs)
};

constexpr int CELL_THREADS = 512;
#ifndef CELL_UNROLL
#define CELL_UNROLL 4   // partners gathered per trip
#endif
constexpr uint32_t CELL_IDX_MASK = (1u << CELL_SPECIES_SHIFT) - 1u;

template <bool ORTHO>
__global__ __launch_bounds__(CELL_THREADS, 8) void rdf_cell_kernel(RdfCellArgs ca)
{
    const RdfFastArgs &fa = ca.f;
    const RdfArgs &a = fa.a;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);          // [npk * nbins]
    unsigned *koff = hist + (size_t)ca.npk * a.nbins;                // [S * S]
    const int tid = threadIdx.x;
    const int S = a.S, nbins = a.nbins;
    // all workgroups of a frame on one XCD (its quantised records stay in that L2).  A workgroup keeps its block of atoms
    // for a chunk of fpc frames (the XCD's own: slot, slot + 8, ...) and flushes its histogram once: flushing per frame was
    // ~2e6 u64 global atomics a frame on configs[4], 854 MB of the kernel's writes (profiles/r03/pmc_cfg4.json)
    unsigned chunk = blockIdx.y, bx = blockIdx.x, xcd = 0, fstep = 1;
    if (fa.xcd_map) {
        const unsigned long long lin = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned long long kk = lin >> 3;
        xcd = (unsigned)(lin & 7ull);
        chunk = (unsigned)(kk / gridDim.x);
        bx = (unsigned)(kk % gridDim.x);
        fstep = 8;
    }
    if ((int)(chunk * (unsigned)ca.fpc * fstep + xcd) >= fa.nf) return;
    for (int k = tid; k < ca.npk * nbins; k += CELL_THREADS) hist[k] = 0u;
    for (int k = tid; k < S * S; k += CELL_THREADS) koff[k] = ca.keyoff[k];
    __syncthreads();
    const int N = (int)a.N;
    const int nx = ca.nx, ny = ca.ny, nz = ca.nz;
    const float half_m_guard = fa.half_m_guard, nb_hi = fa.nb_hi;

    for (int ff = 0; ff < ca.fpc; ff++) {
    const unsigned fl = (chunk * (unsigned)ca.fpc + (unsigned)ff) * fstep + xcd;
    if ((int)fl >= fa.nf) break;
    const int f = fa.f_base + (int)fl;
    const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
    const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
    const int gi = a.n_cells == 1 ? 0 : f;
    const FrameScale *__restrict__ fs = fa.fs + gi;
    const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
    float sc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sc[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->sc[k])));
    const uint32_t *__restrict__ st = ca.start3 + (size_t)fl * ((size_t)nx * ny * nz * S + 1);

    for (int c = 0; c < ca.cpt; c++) {
        const int i = ((int)bx * ca.cpt + c) * CELL_THREADS + tid;
        const bool active = i < N;
        const QAtom own = Qf[active ? i : N - 1];
        const unsigned *krow = koff + (own.idx >> CELL_SPECIES_SHIFT) * S;
        const uint32_t ida = own.idx & CELL_IDX_MASK;
        const int cx = (int)__umulhi(own.ux, (unsigned)nx), cy = (int)__umulhi(own.uy, (unsigned)ny),
                  cz = (int)__umulhi(own.uz, (unsigned)nz);
        // Per-lane trimming of the rows (round 3).  The f32 scales are the lower-triangular factor of the cell's metric, so
        // the Cartesian z of a difference depends on its fractional z only and y on (y, z): from the atom's own place
        // inside its cell, the least |z| and |y| any partner of row (dz, dy) can have leave rho^2 = R^2 - z^2 - y^2 for x,
        // and only the cells of the row within that reach are walked (none if rho^2 < 0).  Bounds are conservative
        // (cell edges widened by 1e-5 of a cell + 1024 units, R by 1e-4 + 2 bins): no pair within rmax is skipped; the
        // visited volume drops from the 5 x 5 x 2.5 cells block to a shell about one cell thick around the half sphere.
        const float uxf = (float)own.ux, uyf = (float)own.uy, uzf = (float)own.uz;
        const float wx = 4294967296.f / (float)nx, wy = 4294967296.f / (float)ny, wz = 4294967296.f / (float)nz;
        const float ex = wx * 1e-5f + 1024.f, ey = wy * 1e-5f + 1024.f, ez = wz * 1e-5f + 1024.f;
        const float Rq = nb_hi * 1.0001f + 2.f;
        const float s_xx = sc[0], s_yx = ORTHO ? 0.f : sc[3], s_yy = ORTHO ? sc[1] : sc[4], s_zx = ORTHO ? 0.f : sc[6],
                    s_zy = ORTHO ? 0.f : sc[7], s_zz = ORTHO ? sc[2] : sc[8];
#pragma unroll 1
        for (int row = 0; row < 13; row++) {
            // rows of the positive half shell
            const int dz = row < 3 ? 0 : (row < 8 ? 1 : 2);
            const int dy = row < 3 ? row : (row < 8 ? row - 5 : row - 10);
            int cz2 = cz + dz, cy2 = cy + dy;
            if (cz2 >= nz) cz2 -= nz;
            if (cy2 >= ny) cy2 -= ny;
            if (cy2 < 0) cy2 += ny;
            const int rowbase = (cz2 * ny + cy2) * nx;
            int xlo = row == 0 ? cx : cx - 2, xhi = cx + 2;
            if (ca.trim) {
                const float zl = (float)(cz + dz) * wz - uzf - ez, zh = zl + wz + 2.f * ez;     // fractional z of the row's partners - mine
                const float yl = (float)(cy + dy) * wy - uyf - ey, yh = yl + wy + 2.f * ey;
                const float dzmin = (zl > 0.f ? zl : (zh < 0.f ? -zh : 0.f)) * s_zz;
                const float yc_lo = yl * s_yy + fminf(zl * s_zy, zh * s_zy), yc_hi = yh * s_yy + fmaxf(zl * s_zy, zh * s_zy);
                const float dymin = yc_lo > 0.f ? yc_lo : (yc_hi < 0.f ? -yc_hi : 0.f);
                const float rem = Rq * Rq - dzmin * dzmin - dymin * dymin;
                if (rem < 0.f) {
                    xhi = xlo - 1;      // nothing of this row is within reach of this atom
                } else {
                    const float rho = __builtin_amdgcn_sqrtf(rem) * 1.0001f + 1.f;
                    const float e_lo = fminf(yl * s_yx, yh * s_yx) + fminf(zl * s_zx, zh * s_zx);
                    const float e_hi = fmaxf(yl * s_yx, yh * s_yx) + fmaxf(zl * s_zx, zh * s_zx);
                    const float fx_lo = (-rho - e_hi) / s_xx, fx_hi = (rho - e_lo) / s_xx;
                    const float inv_wx = (float)nx * (1.f / 4294967296.f);
                    xlo = max(xlo, (int)floorf((uxf + fx_lo - ex) * inv_wx - 1e-3f));
                    xhi = min(xhi, (int)floorf((uxf + fx_hi + ex) * inv_wx + 1e-3f));
                }
            }
            // cells [xlo, xhi] with periodic wrap: at most two runs
            int xa0, xb0, xa1 = 0, xb1 = -1;
            if (xhi < 0) { xa0 = xlo + nx; xb0 = xhi + nx; }                  // (trimmed runs may lie wholly beyond an edge)
            else if (xlo >= nx) { xa0 = xlo - nx; xb0 = xhi - nx; }
            else if (xlo < 0) { xa0 = xlo + nx; xb0 = nx - 1; xa1 = 0; xb1 = xhi; }
            else if (xhi >= nx) { xa0 = xlo; xb0 = nx - 1; xa1 = 0; xb1 = xhi - nx; }
            else { xa0 = xlo; xb0 = xhi; }
            // both runs are looked up before either is walked (dependent global loads)
            int ja0 = 0, ja1 = 0, jb0 = 0, jb1 = 0;
            if (active && xlo <= xhi) {
                ja0 = (int)st[(size_t)(rowbase + xa0) * S];
                ja1 = (int)st[(size_t)(rowbase + xb0 + 1) * S];
                if (row == 0) ja0 = i + 1;                    // own cell: the partners after me
                if (xa1 <= xb1) {
                    jb0 = (int)st[(size_t)(rowbase + xa1) * S];
                    jb1 = (int)st[(size_t)(rowbase + xb1 + 1) * S];
                }
            }
#pragma unroll 1
            for (int seg = 0; seg < 2; seg++) {
                const int j0 = seg == 0 ? ja0 : jb0, j1 = seg == 0 ? ja1 : jb1;
                // several partners per trip, all loaded before any is evaluated (the gathers come from L2)
                for (int j = j0; __any(j < j1); j += CELL_UNROLL) {
                    uint4 qv[CELL_UNROLL];
#pragma unroll
                    for (int u = 0; u < CELL_UNROLL; u++)
                        qv[u] = j < j1 ? *reinterpret_cast<const uint4 *>(Qf + min(j + u, j1 - 1)) : make_uint4(0, 0, 0, 0);
                    // (round 4, measured and rejected: the tile kernel's always-add histogram with trash words per row -- every
                    //  lane adds, only candidates near a bin edge branch: 0.0606 -> 0.0692 ms / frame on configs[4]; three
                    //  quarters of the gathered candidates are out of range and now pay the binning tail too)
#pragma unroll
                    for (int u = 0; u < CELL_UNROLL; u++) {
                        if (j + u < j1) {
                            const uint4 qj = qv[u];
                            const int ix = (int)(qj.x - own.ux), iy = (int)(qj.y - own.uy), iz = (int)(qj.z - own.uz);
                            const float q = fast_q<ORTHO>(sc, ix, iy, iz);
                            if (q < nb_hi) {
                                unsigned *h = hist + krow[qj.w >> CELL_SPECIES_SHIFT];
                                if (fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < half_m_guard) {
                                    atomicAdd(&h[(int)q], 1u);
                                } else {
                                    rdf_pair_refine<ORTHO>(h, fa, fs->sc64, g, q, own.ux, own.uy, own.uz,
                                                           make_uint4(qj.x, qj.y, qj.z, qj.w & CELL_IDX_MASK), p, ida);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    }   // frames of the chunk
    __syncthreads();
    for (int k = tid; k < ca.npk * nbins; k += CELL_THREADS) {
        const unsigned v = hist[k];
        if (v) atomicAdd(&a.U[(size_t)ca.keyU[k / nbins] + (size_t)(k % nbins)], (unsigned long long)v);
    }
}

// --------------------------------------------------------------------------
// Cell-list kernel, one WAVE per centre cell (round 5; the default of the "rdf_cell" family).
//
// rdf_cell_kernel above gives every lane its own centre atom and lets it gather its own partners: the lanes of a wave walk
// runs of different lengths (a wave lasts as long as its longest lane, row by row), every candidate is a 16-byte gather per
// lane, and the kernel runs at ~1e12 candidates/s where the LDS-broadcast tile kernel does 3.6e12 pair evaluations/s.
// Here the roles are swapped and the centre side is made wave-uniform:
//   * a wave owns one CELL of the same grid (>= rmax / 2 thick, ~8 atoms).  The partners of ALL its atoms are the same
//     13 half-shell rows, trimmed ONCE per cell to the cells within reach of the cell (the per-lane trim of the kernel above
//     with the cell's extent in place of the atom's point): up to 26 contiguous index runs, ~650 partners;
//   * the runs are concatenated (lanes 0 .. 12 compute one row each: bounds, table look-ups; a 16-lane scan gives the run
//     offsets, kept in LDS) and the LANES take consecutive partners of the concatenation, 64 at a time, by a five-step
//     binary search over the 32 run offsets: a dense, mostly contiguous load of 64 records, every lane busy but in the
//     last chunk of a cell;
//   * the cell's centre atoms are wave-uniform: their records come through the scalar cache into SGPRs, the pair arithmetic
//     reads them as scalar operands, and the "once" rule needs one compare only in the chunks that hold the cell's own atoms
//     (partner behind the centre in the sorted order; every other row of the half shell counts whole).
// Per centre the wave evaluates the cell's ~650 partners instead of the ~525 the per-lane trim leaves, but every evaluation
// is a dense broadcast one.  Same three-level pair arithmetic, same LDS histogram of every unordered species pair, same frame
// chunks and XCD mapping as the kernel above (RdfCellArgs; cpt = cell groups of 8 cells per workgroup).
#ifndef CW_THREADS_N
#define CW_THREADS_N 512
#endif
#ifndef CW_MIN_WAVES
#define CW_MIN_WAVES 4
#endif
constexpr int CW_THREADS = CW_THREADS_N;
constexpr int CW_WAVES = CW_THREADS / 64;
constexpr int CW_RUNS = 32;          // 13 rows x 2 runs, padded to a power of two with the total
constexpr int CW_QCAP = 126;         // flagged pairs a wave parks per cell (~40 on configs[4]); beyond: refined in place
constexpr int CW_WAVE_WORDS = 2 * CW_RUNS + 2 + 2 * CW_QCAP;     // LDS words per wave behind the histograms

template <bool ORTHO>
__global__ __launch_bounds__(CW_THREADS, CW_MIN_WAVES) void rdf_cellwave_kernel(RdfCellArgs ca)
{
    const RdfFastArgs &fa = ca.f;
    const RdfArgs &a = fa.a;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);          // [npk * nbins]
    unsigned *koff = hist + (size_t)ca.npk * a.nbins;                // [S * S]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.S, nbins = a.nbins;
    // (per-wave tables on an even word: the queue entries are 8-byte pairs)
    unsigned *pre = hist + (((size_t)ca.npk * a.nbins + (size_t)S * S + 1) & ~(size_t)1) + wave * CW_WAVE_WORDS;   // [CW_RUNS] first element of run k
    unsigned *rbase = pre + CW_RUNS;                                 // [CW_RUNS] sorted index of the run's first atom
    unsigned *wq_count = rbase + CW_RUNS;                            // the wave's queue of flagged pairs: entries so far
    uint2 *wq = reinterpret_cast<uint2 *>(wq_count + 2);             // [CW_QCAP] (partner, centre) by sorted index
    unsigned chunk = blockIdx.y, bx = blockIdx.x, xcd = 0, fstep = 1;
    if (fa.xcd_map) {
        const unsigned long long lin = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned long long kk = lin >> 3;
        xcd = (unsigned)(lin & 7ull);
        chunk = (unsigned)(kk / gridDim.x);
        bx = (unsigned)(kk % gridDim.x);
        fstep = 8;
    }
    if ((int)(chunk * (unsigned)ca.fpc * fstep + xcd) >= fa.nf) return;
    for (int k = tid; k < ca.npk * nbins; k += CW_THREADS) hist[k] = 0u;
    for (int k = tid; k < S * S; k += CW_THREADS) koff[k] = ca.keyoff[k];
    __syncthreads();
    const int nx = ca.nx, ny = ca.ny, nz = ca.nz;
    const int ncell = nx * ny * nz;
    const float half_m_guard = fa.half_m_guard, nb_hi = fa.nb_hi;

    for (int ff = 0; ff < ca.fpc; ff++) {
        const unsigned fl = (chunk * (unsigned)ca.fpc + (unsigned)ff) * fstep + xcd;
        if ((int)fl >= fa.nf) break;
        const int f = fa.f_base + (int)fl;
        const double *__restrict__ p = a.pos + (size_t)f * (size_t)a.N * 3;
        const QAtom *__restrict__ Qf = fa.Q + (size_t)fl * (size_t)a.N;
        const int gi = a.n_cells == 1 ? 0 : f;
        const FrameScale *__restrict__ fs = fa.fs + gi;
        const double *__restrict__ g = a.geom + (size_t)gi * GEOM_STRIDE;
        float sc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) sc[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fs->sc[k])));
        const uint32_t *__restrict__ st = ca.start3 + (size_t)fl * ((size_t)ncell * S + 1);
        // the row bounds relative to the centre cell depend on the row and the frame's scales only (every cell has the same
        // extent): lanes 0 .. 12 hold them for the cells of this frame
        int rel_lo = -2, rel_hi = 2;
        const int row = lane < 13 ? lane : 12;
        const int rdz = row < 3 ? 0 : (row < 8 ? 1 : 2);
        const int rdy = row < 3 ? row : (row < 8 ? row - 5 : row - 10);
        if (row == 0) rel_lo = 0;
        if (ca.trim) {
            const float wx = 4294967296.f / (float)nx, wy = 4294967296.f / (float)ny, wz = 4294967296.f / (float)nz;
            const float ex = wx * 2e-5f + 2048.f, ey = wy * 2e-5f + 2048.f, ez = wz * 2e-5f + 2048.f;
            const float Rq = nb_hi * 1.0001f + 2.f;
            const float s_xx = sc[0], s_yx = ORTHO ? 0.f : sc[3], s_yy = ORTHO ? sc[1] : sc[4], s_zx = ORTHO ? 0.f : sc[6],
                        s_zy = ORTHO ? 0.f : sc[7], s_zz = ORTHO ? sc[2] : sc[8];
            // differences partner - centre in grid units: the partner anywhere in its cell, the centre anywhere in mine
            const float zl = (float)(rdz - 1) * wz - ez, zh = (float)(rdz + 1) * wz + ez;
            const float yl = (float)(rdy - 1) * wy - ey, yh = (float)(rdy + 1) * wy + ey;
            const float dzmin = (zl > 0.f ? zl : (zh < 0.f ? -zh : 0.f)) * s_zz;
            const float yc_lo = yl * s_yy + fminf(zl * s_zy, zh * s_zy), yc_hi = yh * s_yy + fmaxf(zl * s_zy, zh * s_zy);
            const float dymin = yc_lo > 0.f ? yc_lo : (yc_hi < 0.f ? -yc_hi : 0.f);
            const float rem = Rq * Rq - dzmin * dzmin - dymin * dymin;
            if (rem < 0.f) {
                rel_hi = rel_lo - 1;        // nothing of this row is within reach of any atom of the cell
            } else {
                const float rho = __builtin_amdgcn_sqrtf(rem) * 1.0001f + 1.f;
                const float e_lo = fminf(yl * s_yx, yh * s_yx) + fminf(zl * s_zx, zh * s_zx);
                const float e_hi = fmaxf(yl * s_yx, yh * s_yx) + fmaxf(zl * s_zx, zh * s_zx);
                // partner x - (my cell's lower edge) lies in [fx_lo - ex, wx + fx_hi + ex]
                const float fx_lo = (-rho - e_hi) / s_xx, fx_hi = (rho - e_lo) / s_xx;
                const float inv_wx = (float)nx * (1.f / 4294967296.f);
                rel_lo = max(rel_lo, (int)floorf((fx_lo - ex) * inv_wx - 1e-3f));
                rel_hi = min(rel_hi, (int)floorf((wx + fx_hi + ex) * inv_wx + 1e-3f));
                if (row == 0) rel_lo = 0;
            }
        }

        for (int cc = 0; cc < ca.cpt; cc++) {
            const int cell = (int)((bx * (unsigned)ca.cpt + (unsigned)cc) * CW_WAVES) + wave;      // wave-uniform
            if (cell >= ncell) break;
            const int c_begin = __builtin_amdgcn_readfirstlane((int)st[(size_t)cell * S]);
            const int c_end = __builtin_amdgcn_readfirstlane((int)st[(size_t)(cell + 1) * S]);
            const int nown = c_end - c_begin;
            if (nown <= 0) continue;
            const int cx = cell % nx, cy = (cell / nx) % ny, cz = cell / (nx * ny);
            // ---- the cell's partner runs: one row per lane (lanes 0 .. 12) ----
            int ja0 = 0, ja1 = 0, jb0 = 0, jb1 = 0;
            if (lane < 13) {
                int cz2 = cz + rdz, cy2 = cy + rdy;
                if (cz2 >= nz) cz2 -= nz;
                if (cy2 >= ny) cy2 -= ny;
                if (cy2 < 0) cy2 += ny;
                const int rowbase = (cz2 * ny + cy2) * nx;
                const int xlo = cx + rel_lo, xhi = cx + rel_hi;
                if (xlo <= xhi) {
                    int xa0, xb0, xa1 = 0, xb1 = -1;
                    if (xhi < 0) { xa0 = xlo + nx; xb0 = xhi + nx; }
                    else if (xlo >= nx) { xa0 = xlo - nx; xb0 = xhi - nx; }
                    else if (xlo < 0) { xa0 = xlo + nx; xb0 = nx - 1; xa1 = 0; xb1 = xhi; }
                    else if (xhi >= nx) { xa0 = xlo; xb0 = nx - 1; xa1 = 0; xb1 = xhi - nx; }
                    else { xa0 = xlo; xb0 = xhi; }
                    ja0 = (int)st[(size_t)(rowbase + xa0) * S];
                    ja1 = (int)st[(size_t)(rowbase + xb0 + 1) * S];
                    if (xa1 <= xb1) {
                        jb0 = (int)st[(size_t)(rowbase + xa1) * S];
                        jb1 = (int)st[(size_t)(rowbase + xb1 + 1) * S];
                    }
                }
            }
            const int la = ja1 - ja0, lb = jb1 - jb0;
            int incl = la + lb;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int n = __shfl_up(incl, off, 64);
                if (lane >= off) incl += n;
            }
            const int T = __builtin_amdgcn_readlane(incl, 15);         // (lanes 13 .. 15 add nothing)
            if (lane < 16) {
                const int excl = incl - la - lb;
                pre[2 * lane] = (unsigned)excl;
                pre[2 * lane + 1] = (unsigned)(excl + la);
                rbase[2 * lane] = (unsigned)ja0;
                rbase[2 * lane + 1] = (unsigned)jb0;
            }
            // (the wave reads what its own lanes wrote: LDS operations of one wave complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // row 0's first run starts with the cell's own atoms: elements 0 .. nown - 1 of the concatenation
            auto locate = [&](int e) {          // sorted index of element e of the concatenation (e < T)
                int k = 0;
#pragma unroll
                for (int step = CW_RUNS / 2; step > 0; step >>= 1)
                    if (pre[k + step] <= (unsigned)e) k += step;
                return (int)rbase[k] + (e - (int)pre[k]);
            };
            // a pair whose candidate lies within the guard of a bin edge is parked in the wave's LDS queue (partner and centre by
            // sorted index) and refined at the end of the cell, one pair per lane: refined in line, the ~150-instruction body ran
            // in 40 % of the centre steps for one or two live lanes (17 of 64 lanes are in range, ~3 % of those near an edge)
            if (lane == 0) *wq_count = 0u;
            int jn = lane < T ? locate(lane) : c_begin;
            uint4 qn = *reinterpret_cast<const uint4 *>(Qf + jn);
            for (int cb = 0; cb < T; cb += 64) {
                const int e = cb + lane;
                const bool alive = e < T;
                const int j = jn;
                const uint4 qj = qn;
                if (cb + 64 < T) {              // the next chunk's records are on their way while this one is evaluated
                    jn = e + 64 < T ? locate(e + 64) : c_begin;
                    qn = *reinterpret_cast<const uint4 *>(Qf + jn);
                }
                const unsigned *kp = koff + (qj.w >> CELL_SPECIES_SHIFT) * S;
                // a partner among the cell's own atoms counts for the centres before it; everything else for every centre
                const int e_own = !alive ? -1 : (e < nown ? e : 0x7fffffff);
                // the cell's atoms are sorted by species (the table has an entry per cell and species): one histogram per
                // species segment, its LDS offset read once -- the inner loop then holds no LDS read, so the NEXT centre's
                // record can be on its way through the scalar cache while this one is evaluated
                auto centres = [&](auto own_tag) {
                    constexpr bool OWN = decltype(own_tag)::value;
                    int seg_b = c_begin;
                    for (int s = 0; s < S; s++) {
                        const int seg_e = __builtin_amdgcn_readfirstlane((int)st[(size_t)cell * S + s + 1]);
                        if (seg_e > seg_b) {
                            unsigned *h = hist + kp[s];
                            QAtom own = Qf[seg_b];
                            for (int ci = seg_b; ci < seg_e; ci++) {
                                const QAtom nxt = Qf[min(ci + 1, seg_e - 1)];          // wave-uniform: scalar loads
                                const int ix = (int)(qj.x - own.ux), iy = (int)(qj.y - own.uy), iz = (int)(qj.z - own.uz);
                                const float q = fast_q<ORTHO>(sc, ix, iy, iz);
                                const bool live = OWN ? (e_own > ci - c_begin) : true;
                                if (live && q < nb_hi) {
                                    if (fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < half_m_guard) {
                                        atomicAdd(&h[(int)q], 1u);
                                    } else {
                                        const unsigned slot = atomicAdd(wq_count, 1u);
                                        if (slot < (unsigned)CW_QCAP) wq[slot] = make_uint2((unsigned)j, (unsigned)ci);
                                        else        // queue full: in place
                                            rdf_pair_refine<ORTHO>(h, fa, fs->sc64, g, q, own.ux, own.uy, own.uz,
                                                                   make_uint4(qj.x, qj.y, qj.z, qj.w & CELL_IDX_MASK), p, own.idx & CELL_IDX_MASK);
                                    }
                                }
                                own = nxt;
                            }
                        }
                        seg_b = seg_e;
                    }
                };
                if (cb < nown) centres(std::true_type{});
                else if (alive) centres(std::false_type{});
            }
            // ---- the cell's flagged pairs: one per lane, the candidate evaluated again from the same records ----
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nflag = min((int)*wq_count, CW_QCAP);
            for (int k = lane; k < nflag; k += 64) {
                const uint2 pr = wq[k];
                const uint4 qb = *reinterpret_cast<const uint4 *>(Qf + pr.x);
                const uint4 qa = *reinterpret_cast<const uint4 *>(Qf + pr.y);
                unsigned *h = hist + koff[(qb.w >> CELL_SPECIES_SHIFT) * S + (qa.w >> CELL_SPECIES_SHIFT)];
                const float q = fast_q<ORTHO>(sc, (int)(qb.x - qa.x), (int)(qb.y - qa.y), (int)(qb.z - qa.z));
                rdf_pair_refine<ORTHO>(h, fa, fs->sc64, g, q, qa.x, qa.y, qa.z, make_uint4(qb.x, qb.y, qb.z, qb.w & CELL_IDX_MASK), p,
                                       qa.w & CELL_IDX_MASK);
            }
            __builtin_amdgcn_wave_barrier();     // (the next cell overwrites the run table)
        }
    }   // frames of the chunk
    __syncthreads();
    for (int k = tid; k < ca.npk * nbins; k += CW_THREADS) {
        const unsigned v = hist[k];
        if (v) atomicAdd(&a.U[(size_t)ca.keyU[k / nbins] + (size_t)(k % nbins)], (unsigned long long)v);
    }
}

// hist[a][b][k] += (a == b) ? 2*U[a][a][k] + nsp[a]*selfh[k] : U[min][max][k]
__global__ void rdf_finalize_kernel(const unsigned long long *U, const unsigned long long *selfh,
                                    const long long *nsp, unsigned long long *hist, int S, int nbins)
{
    size_t total = (size_t)S * S * nbins;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(idx % nbins);
        int ab = (int)(idx / nbins);
        int sa = ab / S, sb = ab % S;
        int lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        unsigned long long u = U[((size_t)lo * S + hi) * nbins + k];
        unsigned long long add = (sa == sb) ? 2ull * u + (unsigned long long)nsp[sa] * selfh[k] : u;
        hist[idx] += add;
    }
}

static int rdf_run(amof_ctx *ctx, const amof_traj *t, double rmax, int32_t nbins,
                   unsigned long long *hist_dev, double *volume_sum)
{
    const int S = t->n_species;
    HostGeom geom;
    AMOF_TRY(build_geometry(ctx, t, geom));
    std::vector<double> img;
    std::vector<int32_t> nimg;
    int max_img = 0;
    AMOF_TRY(build_images(ctx, t, geom, rmax, img, nimg, max_img));
    const double dr = rmax / nbins;
    const double rmax2 = rmax * rmax;

    // self-image pairs (i, i, E): position independent, histogrammed on the host
    std::vector<unsigned long long> selfh((size_t)nbins, 0ull);
    for (int64_t f = 0; f < t->n_frames; f++) {
        size_t gi = t->n_cells == 1 ? 0 : (size_t)f;
        for (int m = 0; m < nimg[gi]; m++) {
            const double *E = &img[(gi * max_img + m) * 3];
            double d2 = fma(E[2], E[2], fma(E[1], E[1], E[0] * E[0]));
            if (d2 < rmax2) {
                int b = (int)(sqrt(d2) / dr);
                if (b < nbins) selfh[(size_t)b]++;
            }
        }
    }

    HostTiles tiles;
    build_tiles(t, RDF_TILE, tiles);
    std::vector<int2> pairs;
    for (int i = 0; i < (int)tiles.tiles.size(); i++)
        for (int j = i; j < (int)tiles.tiles.size(); j++) pairs.push_back(make_int2(i, j));

    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    Stager stage;
    AMOF_TRY(stager_begin(ctx, t, true, stage));
    const double *pos_dev = stage.dev;
    void *d_geom, *d_img, *d_nimg, *d_perm, *d_tiles, *d_pairs, *d_U, *d_self, *d_nsp;
    AMOF_TRY(upload(ctx, SLOT_GEOM, geom.rec.data(), geom.rec.size() * sizeof(double), &d_geom));
    AMOF_TRY(upload(ctx, SLOT_IMG, img.data(), img.size() * sizeof(double), &d_img));
    AMOF_TRY(upload(ctx, SLOT_NIMG, nimg.data(), nimg.size() * sizeof(int32_t), &d_nimg));
    AMOF_TRY(upload(ctx, SLOT_PERM, tiles.perm.data(), tiles.perm.size() * sizeof(int32_t), &d_perm));
    AMOF_TRY(upload(ctx, SLOT_TILES, tiles.tiles.data(), tiles.tiles.size() * sizeof(Tile), &d_tiles));
    AMOF_TRY(upload(ctx, SLOT_PAIRS, pairs.data(), pairs.size() * sizeof(int2), &d_pairs));
    AMOF_TRY(upload(ctx, SLOT_SELF, selfh.data(), selfh.size() * sizeof(unsigned long long), &d_self));
    AMOF_TRY(upload(ctx, SLOT_AUX0, tiles.nsp.data(), tiles.nsp.size() * sizeof(int64_t), &d_nsp));
    size_t U_bytes = (size_t)S * S * nbins * sizeof(unsigned long long);
    AMOF_TRY(ensure(ctx, SLOT_HISTU, U_bytes, &d_U));
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_U, 0, U_bytes, ctx->stream));

    if (!pairs.empty() && t->n_frames > 0) {
        RdfArgs a;
        a.pos = pos_dev;
        a.geom = (const double *)d_geom;
        a.img = (const double *)d_img;
        a.nimg = (const int32_t *)d_nimg;
        a.perm = (const int32_t *)d_perm;
        a.tiles = (const Tile *)d_tiles;
        a.pairs = (const int2 *)d_pairs;
        a.U = (unsigned long long *)d_U;
        a.N = t->n_atoms;
        a.F = (int32_t)t->n_frames;
        a.n_cells = (int32_t)t->n_cells;
        a.nbins = nbins;
        a.S = S;
        a.max_img = max_img;
        a.rmax2 = rmax2;
        a.dr = dr;
        const bool extra = max_img > 0;
        const bool ortho = geom.all_ortho;
        auto pick_chunks = [&](int64_t nframes, int32_t &fpc_out, unsigned &chunks_out) {
            // enough workgroups to fill 256 CUs several times over, few enough
            // flushes that the u64 global atomics stay negligible
            int64_t want_chunks = (8 * 2048 + (int64_t)pairs.size() - 1) / (int64_t)pairs.size();
            int64_t fpc = std::max<int64_t>(1, nframes / std::max<int64_t>(1, want_chunks));
            fpc = std::min<int64_t>(fpc, 64);
            int64_t chunks = (nframes + fpc - 1) / fpc;
            if (chunks > 65535) {
                fpc = (nframes + 65534) / 65535;
                chunks = (nframes + fpc - 1) / fpc;
            }
            fpc_out = (int32_t)fpc;
            chunks_out = (unsigned)chunks;
        };

        // ---- fast path: fixed-point minimum image + guarded candidate bins ----
        const int64_t nc = t->n_cells;
        double csum = 0.0;   // largest sum of cell-vector lengths: bounds the fixed-point grid error
        for (int64_t k = 0; k < nc; k++) {
            const double *c = t->cell + 9 * k;
            csum = std::max(csum, sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) +
                                      sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5]) +
                                      sqrt(c[6] * c[6] + c[7] * c[7] + c[8] * c[8]));
        }
        // fractional coordinates are truncated to 2^-32: a pair vector moves by < csum * 2^-32;
        // quant carries a factor 2 of margin (it also covers the f64 rounding of the fold, |s| < 1e4)
        const double quant = csum * (1.0 / 2147483648.0);
        const double guard_m = quant / dr + (double)nbins * 1e-12;
        // f32 candidate: relative error bound of the chain, see fast_guard_rel (amof_internal.h)
        const double guard_f = (double)nbins * fast_guard_rel_rdf(geom, t->cell, nc) + guard_m;
        const char *force = getenv("AMOF_RDF_KERNEL");
        bool fast = !extra && t->pbc[0] && t->pbc[1] && t->pbc[2] && nbins <= AMOF_MAX_LDS_BINS - 5120 &&
                    guard_f < 0.25 && !(force && strcmp(force, "v1") == 0);
        if (fast && !ortho) {
            // At |s_k| = 1/2 the wrapped fixed-point image and the canonical rint() image can be different
            // lattice images of unequal length in a sheared cell (equal in a diagonal one): keep both
            // decisively out of range, i.e. the cutoff clear of every half height by more than the guards.
            for (int64_t k = 0; k < nc && fast; k++)
                for (int x = 0; x < 3; x++)
                    if (rmax * (1.0 + 4.0 * guard_f / (double)nbins + 1e-6) >= 0.5 * geom.rec[(size_t)k * GEOM_STRIDE + 18 + x])
                        fast = false;
        }
        // Image-aware variant of the tile kernel: the cutoff reaches beyond half a cell height (further periodic
        // images count) or comes too close to it.  Pairs whose fractional difference lies within reach of a cell
        // face are evaluated canonically (base + every listed image); all others can only have their base image
        // in range, and that base is unambiguous.  near_thr[cell][axis]: fixed-point threshold on |i_k|.
        bool fast_img = false;
        double img_share = 0.0;
        std::vector<uint32_t> near_thr;
        if (!fast && t->pbc[0] && t->pbc[1] && t->pbc[2] && nbins <= AMOF_MAX_LDS_BINS - 5120 && guard_f < 0.25 &&
            !(force && strcmp(force, "v1") == 0) && max_img <= 124 && !getenv("AMOF_RDF_NOIMG")) {
            fast_img = true;
            near_thr.assign((size_t)nc * 3, 0x7fffffffu);
            for (int64_t k = 0; k < nc && fast_img; k++) {
                double share = 0.0;      // expected share of the pairs that go the canonical way
                for (int x = 0; x < 3; x++) {
                    const double h = geom.rec[(size_t)k * GEOM_STRIDE + 18 + x];
                    share += 2.0 * std::max(0.0, rmax / h - 0.5);
                }
                // the parking queue holds 1024 of a step's 65 536 pairs; beyond that the exact kernels are the better choice
                if (share > 0.03) fast_img = false;
                img_share = std::max(img_share, share);
            }
            for (int64_t k = 0; k < nc && fast_img; k++)
                for (int x = 0; x < 3; x++) {
                    const double h = geom.rec[(size_t)k * GEOM_STRIDE + 18 + x];
                    // beyond |s| = 1/2 - tau an image shifted along this axis could come within the cutoff
                    const double tau = rmax * (1.0 + 4.0 * guard_f / (double)nbins + 1e-6) / h - 0.5 + 1e-9;
                    const double thr = 0.5 - std::max(tau, 0.0) - 1e-9;
                    if (tau <= 0.0) {
                        // clear of this half height by more than the guards: only an exact tie is ambiguous
                        near_thr[(size_t)k * 3 + x] = 0x7fffffffu;
                    } else if (thr < 0.3) {
                        fast_img = false;      // two fifths of the pairs or more would go the canonical way
                    } else {
                        near_thr[(size_t)k * 3 + x] = (uint32_t)floor(thr * 4294967296.0) - 2u;
                    }
                }
        }
        if (fast && getenv("AMOF_RDF_FORCE_IMG")) {   // tests / measurements: the image-aware variant on a plain case
            fast = false;
            fast_img = true;
            near_thr.assign((size_t)nc * 3, 0x7fffffffu);
        }
        // ---- TRI: general cells in the orthogonalised lattice frame (see fast_quad_tri) ----
        // Stored order (x, y, z = slab axis): of the two orders of the other axes the one that leaves the x wrap the
        // larger slack.  Conditions per cell, with R = rmax (1 + guards) and the lower factor L in Angstrom:
        //   X: R / L00 + |L10| / (2 L00) < 1/2 - 1e-6     (the x wrap, decided without the c10 iy term, cannot lose an in-range image)
        //   Y: tau_y = R / L11 - 1/2 <= 0: unique; else pairs with |iy| > 1/2 - tau_y are flagged near (<= 1.5 % of them)
        //   Z: likewise with L22 (= the slab axis's perpendicular height)
        bool tri = false;
        int tri_code = 0, tri_ax0 = -1, tri_ax1 = -1, tri_axis = 0;       // code: near tests (0 none, 1 slow path only, 2 fast path y, 3 y + z, 4 y with its twin image) + 5 x (x wrap with the y term)
        double tri_share = 0.0, tri_l10_bins = 0.0, tri_c10 = 0.0, tri_tau = 0.0;
        std::vector<double> tri_fold;       // [nc][2] kx, ky
        std::vector<double> tri_rec;        // [nc][9] L00 L10 L11 L22 (bins per 2^-32), thr_y (units), thr_z (bins), kx, ky, thr_x (units)
        const double two32_ = 1.0 / 4294967296.0;
        if (!ortho && max_img <= 124 && t->pbc[0] && t->pbc[1] && t->pbc[2] && nbins <= AMOF_MAX_LDS_BINS - 5120 &&
            guard_f < 0.25 && !(force && strcmp(force, "v1") == 0) && !getenv("AMOF_RDF_NOTRI") && !getenv("AMOF_RDF_FORCE_IMG")) {
            double hmin3[3] = {1e300, 1e300, 1e300};
            for (int64_t k = 0; k < nc; k++)
                for (int x = 0; x < 3; x++) hmin3[x] = std::min(hmin3[x], geom.rec[(size_t)k * GEOM_STRIDE + 18 + x]);
            tri_axis = 0;
            for (int x = 1; x < 3; x++)
                if (hmin3[x] > hmin3[tri_axis]) tri_axis = x;
            const double R = rmax * (1.0 + 4.0 * guard_f / (double)nbins + 1e-6);
            double best_cost = 1e300;
            for (int sw = 0; sw < 2; sw++) {
                const int a0 = sw ? (tri_axis + 2) % 3 : (tri_axis + 1) % 3, a1 = sw ? (tri_axis + 1) % 3 : (tri_axis + 2) % 3;
                const int ordt[3] = {a0, a1, tri_axis};
                bool ok = true;
                double slack = 1e300, share = 0.0, l10b = 0.0, c10max = 0.0, tau_max = 0.0;
                int near = 0;
                bool twin = false, xw_wraps = false, twin_wraps = false, twin_short = false, half_p = true, half_m = true;
                std::vector<double> fold((size_t)nc * 2), rec((size_t)nc * 9);
                for (int64_t k = 0; k < nc && ok; k++) {
                    const double *c = t->cell + 9 * k;
                    double rows[9], L[9];
                    for (int q = 0; q < 3; q++)
                        for (int x = 0; x < 3; x++) rows[3 * q + x] = c[3 * ordt[q] + x];
                    lower_factor(rows, L);
                    if (!(L[0] > 0.0 && L[4] > 0.0 && L[8] > 0.0)) { ok = false; break; }
                    // (no slack: the x wrap takes the c10 iy term along, XW -- two instructions more per pair)
                    slack = std::min(slack, 0.5 - 1e-6 - (R / L[0] + 0.5 * fabs(L[3]) / L[0]));
                    c10max = std::max(c10max, fabs(L[3]) / L[0]);
                    double tau_y = R / L[4] - 0.5 + 1e-9, tau_z = R / L[8] - 0.5 + 1e-9;
                    // (a second image along y up to 9 % of the pairs either side -- hexagonal cells: 7.7 % -- is evaluated by the
                    //  slow path itself, near mode 4; along z only what the canonical queue can take)
                    if (tau_y > 0.09 || tau_z > 0.0075) { ok = false; break; }
                    if (tau_y > 0.0075) twin = true;
                    // (near mode 4 counts the nearer of two candidates that differ by a lattice vector +-(B - k A), x wrapped:
                    //  only where every such vector is at least 2 rmax long can the other one never be in range as well)
                    {
                        const double xr = L[3] - L[0] * rint(L[3] / L[0]);
                        if (sqrt(L[4] * L[4] + xr * xr) * (1.0 + 1e-12) < 2.0 * rmax) twin_short = true;
                    }
                    tau_max = std::max(tau_max, std::max(tau_y, 0.0));
                    // The x wrap of XW and of the twin image forms (int)(fy * c10), fy = the y difference in units of 2^-32 of the
                    // cell: modular arithmetic only while |fy c10| < 2^31 (the conversion saturates beyond).  In range means
                    // |fy| <= R / L11 cells (the twin: 1/2 + tau_y), so a skewed, non-reduced cell with |L10| >~ L00 cannot take
                    // these variants: rdf_tile_img / rdf_exact answer it.
                    if (fabs(L[3]) / L[0] * (R / L[4] + 1e-3) >= 0.5 - 1e-3) xw_wraps = true;
                    if (fabs(L[3]) / L[0] * (0.5 + std::max(tau_y, 0.0) + 1e-3) >= 0.5 - 1e-3) twin_wraps = true;
                    // A second image along y (z) can only be in range when L11 / 2 < R0 (canonical rmax with rounding slack);
                    // then its in-plane components are below rho = sqrt(R0^2 - (L/2)^2), the evaluated image's differ from them
                    // by at most the lattice offsets, so it lies between L - R0 and sqrt(D2max) from the origin.  When that
                    // whole interval is within g_m / 2 of the cutoff the pair is flagged by the guard band of the last bin
                    // edge anyway: no compare in the fast path (the slow path tests, and parks it).
                    const double R0 = rmax * (1.0 + 1e-12), gband = 0.5 * (2.0 * quant / dr + (double)nbins * 1e-12);
                    auto covered = [&](double Lk, double off_a, double off_b) {
                        const double rho = sqrt(std::max(0.0, R0 * R0 - 0.25 * Lk * Lk));
                        const double d2max = 0.25 * Lk * Lk + (rho + off_a) * (rho + off_a) + (rho + off_b) * (rho + off_b);
                        return (Lk - R0) / dr >= (double)nbins - gband && sqrt(d2max) / dr <= (double)nbins + gband;
                    };
                    if (0.5 * L[4] >= R0) tau_y = -1.0;       // no second image along y at all
                    else near = std::max(near, covered(L[4], fabs(L[3]), 0.0) ? 1 : 2);
                    if (0.5 * L[8] >= R0) tau_z = -1.0;
                    else near = std::max(near, covered(L[8], fabs(L[7]), fabs(L[6]) + fabs(L[3])) ? 1 : 3);
                    // x: |A| >= 2 rmax whenever rmax is the reference's half shortest length; a larger rmax (the C ABI
                    // takes any) would need a near test on x in the fast path: not this variant
                    double tau_x = R / L[0] - 0.5 + 1e-9;
                    if (0.5 * L[0] >= R0) tau_x = -1.0;
                    else if (covered(L[0], 0.0, 0.0)) near = std::max(near, 1);
                    else { ok = false; break; }
                    share = std::max(share, tau_y > 0.0075 ? 0.012 : 2.0 * std::max(tau_y, 0.0) + 2.0 * std::max(tau_z, 0.0));
                    l10b = std::max(l10b, fabs(L[3]) / dr);
                    const double c10 = L[3] / L[0], r20 = L[6] / L[0], r21 = L[7] / L[4];
                    // (c10 = +-1/2 to 2^-33: the exact-half x wrap of near mode 4, tri_q_twin<HALF>, is then right to one grid unit)
                    if (fabs(c10 - 0.5) > 1e-10) half_p = false;
                    if (fabs(c10 + 0.5) > 1e-10) half_m = false;
                    fold[(size_t)k * 2] = r20 - c10 * r21;
                    fold[(size_t)k * 2 + 1] = r21;
                    double *r = &rec[(size_t)k * 9];
                    r[0] = L[0] * two32_ / dr; r[1] = L[3] * two32_ / dr; r[2] = L[4] * two32_ / dr; r[3] = L[8] * two32_ / dr;
                    // thresholds with room for the f32 conversions / coordinates of the fast path (flag a few more, never fewer)
                    r[4] = tau_y > 0.0 ? (0.5 - tau_y) * 4294967296.0 * (1.0 - 1e-6) - 8.0 : INFINITY;
                    r[5] = tau_z > 0.0 ? (L[8] - R) / dr * (1.0 - 1e-6) - 0.02 : INFINITY;
                    r[6] = fold[(size_t)k * 2]; r[7] = fold[(size_t)k * 2 + 1];
                    r[8] = tau_x > 0.0 ? (0.5 - tau_x) * 4294967296.0 * (1.0 - 1e-6) - 1024.0 : INFINITY;     // (f32 sum: 2^-23 of 2^31)
                }
                // cheaper order first: a near test costs a compare per pair, the x wrap two instructions in the chain
                // (measured: + 8 % per compare, + 24 % for the wrap, profiles/r04/tri_experiments.txt)
                if (twin) {
                    if (near == 3) ok = false;      // (a common twin along y AND near tests along z: the image-aware / exact kernels)
                    if (twin_wraps || twin_short) ok = false;
                    near = 4;
                }
                if (!(slack > 0.0) && xw_wraps) ok = false;
                int code = near + (slack > 0.0 ? 0 : 5);
                if (near == 4 && (half_p || half_m) && !getenv("AMOF_RDF_NOHALF")) code = half_p ? 10 : 11;
                const double cost = (near == 0 ? 0.0 : near == 1 ? 0.5 : near == 4 ? 6.0 : (double)(near - 1) * 1.5) +
                                    (slack > 0.0 ? 0.0 : 2.5);
                if (ok && (!tri || cost < best_cost)) {
                    best_cost = cost;
                    tri = true; tri_code = code; tri_ax0 = a0; tri_ax1 = a1; tri_share = share; tri_l10_bins = l10b; tri_c10 = c10max;
                    tri_tau = tau_max;
                    tri_fold.swap(fold); tri_rec.swap(rec);
                }
            }
        }
        if (tri) {      // (the guard of its candidate at the widest reach must leave room between the bin edges)
            double hb = 0.0;
            for (int64_t k = 0; k < nc; k++) hb = std::max(hb, geom.rec[(size_t)k * GEOM_STRIDE + 18 + tri_axis] / dr);
            if (!(fast_guard_tri(nbins, hb, 0.5, tri_l10_bins) * (1.0 + 4.0 * tri_tau) + 2.0 * quant / dr +
                  (1.0 + 256.0 * tri_c10) * 1.5 * csum * two32_ / dr + (double)nbins * 1e-12 < 0.25)) tri = false;
        }
        if (tri && getenv("AMOF_RDF_DEBUG"))
            fprintf(stderr, "rdf_tile_tri: code %d (near mode %d, x wrap %d) axes (%d, %d | %d) parked share %.4f tau %.5f c10 %.5f\n", tri_code,
                    tri_code >= 10 ? 4 : tri_code % 5, tri_code >= 10 ? 2 : tri_code / 5, tri_ax0, tri_ax1, tri_axis, tri_share, tri_tau, tri_c10);
        bool done = false;
        const bool fast_plain = fast;      // the cell-list / range kernels below rest on the plain criterion (cutoff clear of every half height)
        if (tri) { fast = true; fast_img = false; }
        if (fast || fast_img) {
            HostTiles ftiles;
            build_tiles(t, FAST_TILE, ftiles, FAST_SUB);
            std::vector<int2> fpairs;
            for (int i = 0; i < (int)ftiles.tiles.size(); i++)
                for (int j = i; j < (int)ftiles.tiles.size(); j++) fpairs.push_back(make_int2(i, j));
            // heavy tile pairs first: the drain tail of the launch is filled with the light ones
            {
                auto cost = [&](const int2 &pr) {
                    const double c = (double)ftiles.tiles[pr.x].count * (double)ftiles.tiles[pr.y].count;
                    return pr.x == pr.y ? 0.5 * c : c;
                };
                std::stable_sort(fpairs.begin(), fpairs.end(), [&](const int2 &u, const int2 &v) { return cost(u) > cost(v); });
            }
            void *d_ftiles, *d_fpairs;
            AMOF_TRY(upload(ctx, SLOT_AUX2, ftiles.tiles.data(), ftiles.tiles.size() * sizeof(Tile), &d_ftiles));
            AMOF_TRY(upload(ctx, SLOT_AUX3, fpairs.data(), fpairs.size() * sizeof(int2), &d_fpairs));
            // slab axis = the axis whose smallest perpendicular height over the trajectory is largest;
            // culling only pays when 2 rmax < h
            int axis = 0;
            double hmin[3] = {1e300, 1e300, 1e300};
            for (int64_t k = 0; k < nc; k++)
                for (int x = 0; x < 3; x++) hmin[x] = std::min(hmin[x], geom.rec[(size_t)k * GEOM_STRIDE + 18 + x]);
            for (int x = 1; x < 3; x++)
                if (hmin[x] > hmin[axis]) axis = x;
            const char *nocull = getenv("AMOF_RDF_NOCULL");
            const bool cull = !(nocull && nocull[0] == '1') && 2.0 * rmax * 1.05 < hmin[axis];
            std::vector<int64_t> sp_first(S + 1, 0);
            for (int sidx = 0; sidx < S; sidx++) sp_first[sidx + 1] = sp_first[sidx] + ftiles.nsp[sidx];
            void *d_spfirst;
            AMOF_TRY(upload(ctx, SLOT_AUX4, sp_first.data(), sp_first.size() * sizeof(int64_t), &d_spfirst));
            // per-cell records; fixed-point components are stored in the order (ax0, ax1, axis)
            std::vector<FrameScale> fsv((size_t)nc);
            const int ord[3] = {(axis + 1) % 3, (axis + 2) % 3, axis};
            const double two32 = 1.0 / 4294967296.0;
            for (int64_t k = 0; k < nc; k++) {
                const double *c = t->cell + 9 * k;
                FrameScale &r = fsv[(size_t)k];
                for (int q = 0; q < 9; q++) r.sc64[q] = 0.0;
                if (ortho) {
                    for (int q = 0; q < 3; q++) r.sc64[q] = c[4 * ord[q]] * two32 / dr;
                } else {
                    double rows[9];
                    for (int q = 0; q < 3; q++)
                        for (int x = 0; x < 3; x++) rows[3 * q + x] = c[3 * ord[q] + x] * two32 / dr;
                    lower_factor(rows, r.sc64);
                }
                for (int q = 0; q < 9; q++) r.sc[q] = (float)r.sc64[q];
                if (ortho)
                    for (int q = 0; q < 3; q++) r.sc[3 + q] = (float)(r.sc64[q] * r.sc64[q]);
                // a block is skipped when the slab gap alone exceeds rmax (1e-6 relative and 4 grid units of slack)
                const double hax = geom.rec[(size_t)k * GEOM_STRIDE + 18 + axis];
                r.cull_gap = cull ? (uint32_t)std::min(4294967295.0, ceil(rmax / hax * 4294967296.0 * (1.0 + 1e-6)) + 4.0) : 0u;
                for (int q = 0; q < 3; q++) r.near_t[q] = fast_img ? near_thr[(size_t)k * 3 + ord[q]] : 0x7fffffffu;
                r._pad = 0u;
            }
            void *d_fs;
            AMOF_TRY(upload(ctx, SLOT_AUX5, fsv.data(), fsv.size() * sizeof(FrameScale), &d_fs));
            RdfFastArgs fa;
            fa.a = a;
            fa.a.tiles = (const Tile *)d_ftiles;
            fa.a.pairs = (const int2 *)d_fpairs;
            fa.fs = (const FrameScale *)d_fs;
            // thresholds in f32 with directed rounding, so that the f64 bound g_f is never undercut
            {
                const double hi = (double)nbins + guard_f, half = 0.5 - guard_f;
                float fhi = (float)hi, fhalf = (float)half;
                if ((double)fhi < hi) fhi = nextafterf(fhi, INFINITY);
                if ((double)fhalf > half) fhalf = nextafterf(fhalf, -INFINITY);
                fa.nb_hi = fhi;
                fa.half_m_guard = fhalf;
            }
            fa.guard64 = guard_m;
            fa.guard64_2 = 2.0 * guard_m * (1.0 + 1e-9);
            fa.guard64_sq = guard_m * guard_m * (1.0 + 1e-9);
            // Diagonal cells: the tile kernel takes the slab-axis difference from f32 coordinates (ZF, see fast_q_zf) --
            // with slab culling for every visited partner, without it for all but a band around the sub-tile's antipode;
            // its candidate carries one more absolute term, hence its own (slightly wider) guard.
            const char *nozf = getenv("AMOF_RDF_NOZF");
            bool use_zf = fast && !fast_img && ortho && !(nozf && nozf[0] == '1');
            float zf_nb_hi = fa.nb_hi, zf_half_m_guard = fa.half_m_guard;
            if (use_zf) {
                double hb = 0.0, gfrac = 0.0;
                for (int64_t k = 0; k < nc; k++) {
                    hb = std::max(hb, geom.rec[(size_t)k * GEOM_STRIDE + 18 + axis] / dr);
                    gfrac = std::max(gfrac, cull ? (double)fsv[(size_t)k].cull_gap / 4294967296.0 : 0.5);   // (no culling: reach 2^31)
                }
                const double guard_zf = fast_guard_zf(nbins, hb, gfrac) + guard_m;
                if (!(guard_zf < 0.25)) use_zf = false;
                const double hi = (double)nbins + guard_zf, half = 0.5 - guard_zf;
                float fhi = (float)hi, fhalf = (float)half;
                if ((double)fhi < hi) fhi = nextafterf(fhi, INFINITY);
                if ((double)fhalf > half) fhalf = nextafterf(fhalf, -INFINITY);
                zf_nb_hi = fhi;
                zf_half_m_guard = fhalf;
            }
            // ---- 3-D cell list for cutoffs far below the cell size (cell kernel) ----
            bool cell_taken = false;
            {
                int nk[3];
                bool cell_ok = fast_plain && S <= 16 && t->n_atoms < (1ll << CELL_SPECIES_SHIFT) && t->n_atoms >= 64 &&
                               !getenv("AMOF_RDF_NOCELL");
                for (int x = 0; x < 3; x++) {
                    nk[x] = (int)std::min(1024.0, floor(hmin[x] / (0.5 * rmax * (1.0 + 1e-5))));
                    if (nk[x] < 5) cell_ok = false;
                }
                const int npk = S * (S + 1) / 2;
                const size_t lds3 = ((size_t)npk * nbins + (size_t)S * S) * sizeof(unsigned);
                if (lds3 > 96 * 1024) cell_ok = false;
                if (cell_ok) {
                    // no point in cells emptier than ~2 atoms: thicker cells are still correct (reach stays 2)
                    while ((int64_t)nk[0] * nk[1] * nk[2] > std::max<int64_t>(125, t->n_atoms / 2)) {
                        int big = 0;
                        for (int x = 1; x < 3; x++)
                            if (nk[x] > nk[big]) big = x;
                        if (nk[big] <= 5) break;
                        nk[big]--;
                    }
                    // visited share of all pairs: half shell of 5x5x5 cells (lane utilisation ~0.8) vs the slab list's
                    // f1c / 2; a gathered pair costs ~1.8x a broadcast one and the sort is a fixed cost per frame
                    // (crossovers measured with profiles/tools/sweep_cell.py)
                    const double f3 = 62.5 / ((double)nk[0] * nk[1] * nk[2]) / 0.8;
                    const double f1c = std::min(1.0, 2.0 * rmax / hmin[axis] + 2.0 / 256 + 0.02);
                    if (!(f3 < 0.32 * f1c && t->n_atoms >= 4000) && !getenv("AMOF_RDF_FORCE_CELL")) cell_ok = false;
                }
                if (cell_ok) {
                    const int nkeys = nk[0] * nk[1] * nk[2] * S;
                    AMOF_TRY(stager_need(stage, t->n_frames));
                    std::vector<FrameScale> fs3((size_t)nc);
                    for (int64_t k = 0; k < nc; k++) {
                        const double *c = t->cell + 9 * k;
                        FrameScale &r = fs3[(size_t)k];
                        for (int q = 0; q < 9; q++) r.sc64[q] = 0.0;
                        if (ortho) {
                            for (int q = 0; q < 3; q++) r.sc64[q] = c[4 * q] * two32 / dr;
                        } else {
                            double rows[9];
                            for (int q = 0; q < 9; q++) rows[q] = c[q] * two32 / dr;
                            lower_factor(rows, r.sc64);
                        }
                        for (int q = 0; q < 9; q++) r.sc[q] = (float)r.sc64[q];
                        if (ortho)
                            for (int q = 0; q < 3; q++) r.sc[3 + q] = (float)(r.sc64[q] * r.sc64[q]);
                        r.cull_gap = 0u;
                        r.near_t[0] = r.near_t[1] = r.near_t[2] = 0x7fffffffu;
                        r._pad = 0u;
                    }
                    std::vector<uint32_t> ktab((size_t)S * S + npk);
                    {
                        int key = 0;
                        for (int lo = 0; lo < S; lo++)
                            for (int hi = lo; hi < S; hi++, key++) {
                                ktab[(size_t)lo * S + hi] = ktab[(size_t)hi * S + lo] = (uint32_t)(key * nbins);
                                ktab[(size_t)S * S + key] = (uint32_t)(((size_t)lo * S + hi) * nbins);
                            }
                    }
                    void *d_fs3, *d_ktab, *d_spec, *d_Q3, *d_start3, *d_keys, *d_cursor, *d_flag3;
                    AMOF_TRY(upload(ctx, SLOT_AUX5, fs3.data(), fs3.size() * sizeof(FrameScale), &d_fs3));
                    AMOF_TRY(upload(ctx, SLOT_AUX8, ktab.data(), ktab.size() * sizeof(uint32_t), &d_ktab));
                    AMOF_TRY(upload(ctx, SLOT_SPEC, t->species, (size_t)t->n_atoms * sizeof(int32_t), &d_spec));
                    const size_t per_frame = (size_t)t->n_atoms * (sizeof(QAtom) + sizeof(uint32_t)) +
                                             (size_t)(2 * nkeys + 1) * sizeof(uint32_t);
                    int64_t FB3 = std::max<int64_t>(1, (int64_t)(1ll << 30) / (int64_t)per_frame);
                    FB3 = std::min<int64_t>(std::min<int64_t>(FB3, 32768), t->n_frames);
                    // (tests cross a frame-batch boundary without gigabytes of frames)
                    if (const char *be = getenv("AMOF_RDF_BATCH")) FB3 = std::min<int64_t>(FB3, std::max(1, atoi(be)));
                    AMOF_TRY(ensure(ctx, SLOT_AUX1, (size_t)FB3 * t->n_atoms * sizeof(QAtom), &d_Q3));
                    AMOF_TRY(ensure(ctx, SLOT_AUX7, (size_t)FB3 * (nkeys + 1) * sizeof(uint32_t), &d_start3));
                    AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)FB3 * t->n_atoms * sizeof(uint32_t), &d_keys));
                    AMOF_TRY(ensure(ctx, SLOT_AUX9, (size_t)FB3 * nkeys * sizeof(uint32_t), &d_cursor));
                    AMOF_TRY(ensure(ctx, SLOT_FLAGS, sizeof(int32_t), &d_flag3));
                    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_flag3, 0, sizeof(int32_t), ctx->stream));
                    RdfCellArgs ca;
                    ca.f = fa;
                    ca.f.fs = (const FrameScale *)d_fs3;
                    ca.f.Q = (const QAtom *)d_Q3;
                    ca.start3 = (const uint32_t *)d_start3;
                    ca.keyoff = (const uint32_t *)d_ktab;
                    ca.keyU = (const uint32_t *)d_ktab + (size_t)S * S;
                    ca.nx = nk[0]; ca.ny = nk[1]; ca.nz = nk[2];
                    ca.npk = npk;
                    ca.cpt = 1;
                    ca.trim = getenv("AMOF_RDF_NOTRIM") ? 0 : 1;
                    // round 5: one wave per cell (rdf_cellwave_kernel) unless AMOF_RDF_CELL_GATHER=1 asks for the per-lane gather form
                    const bool wavecell = !getenv("AMOF_RDF_CELL_GATHER");
                    const int64_t ncell3 = (int64_t)nk[0] * nk[1] * nk[2];
                    const size_t lds3w = lds3 + ((size_t)CW_WAVES * CW_WAVE_WORDS + 1) * sizeof(unsigned);
                    unsigned gx = (unsigned)((t->n_atoms + (int64_t)CELL_THREADS * ca.cpt - 1) / ((int64_t)CELL_THREADS * ca.cpt));
                    if (wavecell) gx = (unsigned)((ncell3 + CW_WAVES - 1) / CW_WAVES);
                    int64_t launches = 0;
                    for (int64_t fb = 0; fb < t->n_frames; fb += FB3) {
                        const int64_t nf = std::min<int64_t>(FB3, t->n_frames - fb);
                        AMOF_TRY(launch_cell_sort(ctx, pos_dev, (const double *)d_geom, (int)t->n_cells, (const int32_t *)d_spec, S,
                                                  t->n_atoms, (int)fb, (int)nf, nk[0], nk[1], nk[2], (QAtom *)d_Q3,
                                                  (uint32_t *)d_start3, (uint32_t *)d_keys, (uint32_t *)d_cursor,
                                                  (int32_t *)d_flag3));
                        ca.f.f_base = (int32_t)fb;
                        ca.f.nf = (int32_t)nf;
                        ca.f.xcd_map = nf >= 16 ? 1 : 0;
                        ca.frames_grid = (int32_t)(ca.f.xcd_map ? (nf + 7) / 8 * 8 : nf);
                        // frames per workgroup: >= ~8k workgroups per launch (32 per CU), at most 8 frames (the u32 counters of
                        // a chunk: 8 frames x 512 atoms x ~300 partners stay far below 2^32)
                        {
                            const int64_t slots = ca.f.xcd_map ? ca.frames_grid / 8 : ca.frames_grid;       // per XCD
                            int64_t fpc = std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)gx * ca.frames_grid / 8192));
                            if (const char *fe = getenv("AMOF_RDF_CELL_FPC")) fpc = std::max(1, atoi(fe));      // experiments / tests
                            fpc = std::min<int64_t>(fpc, slots);
                            ca.fpc = (int32_t)fpc;
                            const int64_t chunks = (slots + fpc - 1) / fpc;
                            ca.frames_grid = (int32_t)(ca.f.xcd_map ? chunks * 8 : chunks);
                        }
                        if (wavecell) {
                            // cell groups per workgroup: the histogram flush (npk x nbins u64 atomics) is paid per workgroup and
                            // chunk, a cell and frame bring ~1e3 pairs -- keep >= ~8k workgroups per launch, at most 8 groups
                            const int64_t groups = (ncell3 + CW_WAVES - 1) / CW_WAVES;
                            int64_t cpw = std::max<int64_t>(1, std::min<int64_t>(8, groups * ca.frames_grid / 8192));
                            if (const char *ce = getenv("AMOF_RDF_CELL_CPW")) cpw = std::max(1, atoi(ce));            // experiments / tests
                            cpw = std::min<int64_t>(cpw, groups);
                            ca.cpt = (int32_t)cpw;
                            gx = (unsigned)((groups + cpw - 1) / cpw);
                        }
                        dim3 grid(gx, (unsigned)ca.frames_grid);
                        if (launches == 0) timing_dom_begin(ctx, "rdf_cell");
                        if (wavecell) {
                            hipError_t e = ortho ? allow_max_lds((const void *)rdf_cellwave_kernel<true>)
                                                 : allow_max_lds((const void *)rdf_cellwave_kernel<false>);
                            AMOF_HIP_TRY(ctx, e);
                            if (ortho) hipLaunchKernelGGL(rdf_cellwave_kernel<true>, grid, dim3(CW_THREADS), lds3w, ctx->stream, ca);
                            else hipLaunchKernelGGL(rdf_cellwave_kernel<false>, grid, dim3(CW_THREADS), lds3w, ctx->stream, ca);
                        } else {
                            hipError_t e = ortho ? allow_max_lds((const void *)rdf_cell_kernel<true>)
                                                 : allow_max_lds((const void *)rdf_cell_kernel<false>);
                            AMOF_HIP_TRY(ctx, e);
                            if (ortho) hipLaunchKernelGGL(rdf_cell_kernel<true>, grid, dim3(CELL_THREADS), lds3, ctx->stream, ca);
                            else hipLaunchKernelGGL(rdf_cell_kernel<false>, grid, dim3(CELL_THREADS), lds3, ctx->stream, ca);
                        }
                        AMOF_HIP_TRY(ctx, hipGetLastError());
                        launches++;
                    }
                    timing_dom_end(ctx, launches);
                    int32_t flag3 = 0;
                    AMOF_TRY(fetch(ctx, &flag3, d_flag3, sizeof flag3));
                    AMOF_HIP_TRY(ctx, sync_stream(ctx));
                    if (flag3) AMOF_HIP_TRY(ctx, hipMemsetAsync(d_U, 0, U_bytes, ctx->stream));   // far-away atoms: exact kernels
                    else done = true;
                    cell_taken = true;    // (either done, or the exact kernels take over)
                }
            }
            // ---- two-level cell list for small cutoffs (range kernel) ----
            int axis_y = (axis + 1) % 3;
            if (hmin[(axis + 2) % 3] > hmin[axis_y]) axis_y = (axis + 2) % 3;
            const int nz2 = (int)std::min(64.0, floor(hmin[axis] / (rmax * (1.0 + 1e-5))));
            bool use_range = false;
            if (fast_plain && nz2 >= 3 && t->n_cells == 1 && !cell_taken && !(getenv("AMOF_RDF_NORANGE"))) {
                // visited share of the partners: 1-D slab list vs (3 slabs) x (y strip + 2 rmax)
                int64_t nmax = 0;
                for (int x = 0; x < S; x++) nmax = std::max<int64_t>(nmax, ftiles.nsp[x]);
                const double strip = std::min(1.0, (double)FAST_SUB * nz2 / std::max<double>(1.0, (double)nmax));
                const double f1 = std::min(1.0, 2.0 * rmax / hmin[axis] + 2.0 / 256 + 0.02);
                const double f2 = std::min(1.0, 3.0 / nz2) * std::min(1.0, 2.0 * rmax / hmin[axis_y] + strip + 2.0 / 256);
                use_range = f2 < 0.6 * f1 || getenv("AMOF_RDF_FORCE_RANGE") != nullptr;   // (forced: tests)
            }
            if (use_range) {
                AMOF_TRY(stager_need(stage, t->n_frames));
                const int gy = (int)ceil(rmax * (1.0 + 1e-5) / hmin[axis_y] * 256.0) + 1;
                // scale records: stored component order is (remaining axis, axis_y, axis)
                const int ax_x = 3 - axis - axis_y;
                const int ord2[3] = {ax_x, axis_y, axis};
                for (int64_t k = 0; k < nc; k++) {
                    const double *c = t->cell + 9 * k;
                    FrameScale &r = fsv[(size_t)k];
                    for (int q = 0; q < 9; q++) r.sc64[q] = 0.0;
                    if (ortho) {
                        for (int q = 0; q < 3; q++) r.sc64[q] = c[4 * ord2[q]] * two32 / dr;
                    } else {
                        double rows[9];
                        for (int q = 0; q < 3; q++)
                            for (int x = 0; x < 3; x++) rows[3 * q + x] = c[3 * ord2[q] + x] * two32 / dr;
                        lower_factor(rows, r.sc64);
                    }
                    for (int q = 0; q < 9; q++) r.sc[q] = (float)r.sc64[q];
                    if (ortho)
                        for (int q = 0; q < 3; q++) r.sc[3 + q] = (float)(r.sc64[q] * r.sc64[q]);
                    r.cull_gap = 0u;
                    r.near_t[0] = r.near_t[1] = r.near_t[2] = 0x7fffffffu;
                    r._pad = 0u;
                }
                AMOF_TRY(upload(ctx, SLOT_AUX5, fsv.data(), fsv.size() * sizeof(FrameScale), &d_fs));
                fa.fs = (const FrameScale *)d_fs;
                std::vector<int4> rwork;
                for (int sa2 = 0; sa2 < S; sa2++)
                    for (int64_t c0 = 0; c0 < ftiles.nsp[sa2]; c0 += FAST_SUB)
                        for (int sb2 = sa2; sb2 < S; sb2++)
                            if (ftiles.nsp[sb2] > 0) rwork.push_back(make_int4(sa2, (int)c0, sb2, 0));
                void *d_rwork, *d_start2, *d_Q2, *d_flag2;
                AMOF_TRY(upload(ctx, SLOT_AUX6, rwork.data(), rwork.size() * sizeof(int4), &d_rwork));
                const size_t per_frame = (size_t)t->n_atoms * sizeof(QAtom) + (size_t)S * (nz2 * 256 + 1) * sizeof(uint32_t);
                int64_t FB2 = std::max<int64_t>(1, (int64_t)(1ll << 30) / (int64_t)per_frame);
                FB2 = std::min<int64_t>(std::min<int64_t>(FB2, 32768), t->n_frames);
                AMOF_TRY(ensure(ctx, SLOT_AUX1, (size_t)FB2 * t->n_atoms * sizeof(QAtom), &d_Q2));
                AMOF_TRY(ensure(ctx, SLOT_AUX7, (size_t)FB2 * S * (nz2 * 256 + 1) * sizeof(uint32_t), &d_start2));
                AMOF_TRY(ensure(ctx, SLOT_FLAGS, sizeof(int32_t), &d_flag2));
                AMOF_HIP_TRY(ctx, hipMemsetAsync(d_flag2, 0, sizeof(int32_t), ctx->stream));
                RdfRangeArgs ra;
                ra.f = fa;
                ra.f.Q = (const QAtom *)d_Q2;
                ra.f.xcd_map = 0;
                ra.start2 = (const uint32_t *)d_start2;
                ra.sp_first = (const int64_t *)d_spfirst;
                ra.work = (const int4 *)d_rwork;
                ra.nz = nz2;
                ra.gy_bins = gy;
                size_t lds2 = FAST_TILE * sizeof(uint4) + (size_t)nbins * sizeof(unsigned);
                int64_t launches = 0;
                for (int64_t fb = 0; fb < t->n_frames && !rwork.empty(); fb += FB2) {
                    const int64_t nf = std::min<int64_t>(FB2, t->n_frames - fb);
                    AMOF_TRY(launch_quantize2(ctx, pos_dev, (const double *)d_geom, (int)t->n_cells, (const int32_t *)d_perm,
                                              (const int64_t *)d_spfirst, S, t->n_atoms, (int)fb, (int)nf, axis, axis_y, nz2,
                                              (QAtom *)d_Q2, (uint32_t *)d_start2, (int32_t *)d_flag2));
                    ra.f.f_base = (int32_t)fb;
                    ra.f.nf = (int32_t)nf;
                    int64_t want_chunks = (8 * 2048 + (int64_t)rwork.size() - 1) / (int64_t)rwork.size();
                    int64_t fpc = std::max<int64_t>(1, nf / std::max<int64_t>(1, want_chunks));
                    fpc = std::min<int64_t>(fpc, 16);
                    int64_t chunks = (nf + fpc - 1) / fpc;
                    ra.f.a.frames_per_chunk = (int32_t)fpc;
                    dim3 grid((unsigned)rwork.size(), (unsigned)chunks);
                    if (launches == 0) timing_dom_begin(ctx, "rdf_range");
                    hipError_t e;
                    if (ortho) {
                        e = allow_max_lds((const void *)rdf_range_kernel_fast<true>);
                        if (e == hipSuccess) hipLaunchKernelGGL(rdf_range_kernel_fast<true>, grid, dim3(FAST_THREADS), lds2, ctx->stream, ra);
                    } else {
                        e = allow_max_lds((const void *)rdf_range_kernel_fast<false>);
                        if (e == hipSuccess) hipLaunchKernelGGL(rdf_range_kernel_fast<false>, grid, dim3(FAST_THREADS), lds2, ctx->stream, ra);
                    }
                    AMOF_HIP_TRY(ctx, e);
                    AMOF_HIP_TRY(ctx, hipGetLastError());
                    launches++;
                }
                timing_dom_end(ctx, launches);
                int32_t flag2 = 0;
                AMOF_TRY(fetch(ctx, &flag2, d_flag2, sizeof flag2));
                AMOF_HIP_TRY(ctx, sync_stream(ctx));
                if (flag2) AMOF_HIP_TRY(ctx, hipMemsetAsync(d_U, 0, U_bytes, ctx->stream));
                else done = true;
            }
            if (!done && !use_range && !cell_taken) {
            int64_t FB = std::max<int64_t>(1, (int64_t)(1ll << 30) / std::max<int64_t>(1, t->n_atoms * 16));
            FB = std::min<int64_t>(std::min<int64_t>(FB, 32768), t->n_frames);
            // host-resident input: batches of 512, 1024, 2048 ... frames (each >= 32 chunks of 16 frames); the copy
            // of batch k+1 overlaps the kernels of batch k, and only the first, small copy is exposed
            int64_t cur = stage.lazy ? std::min<int64_t>(FB, 512) : FB;
            if (const char *be = getenv("AMOF_RDF_BATCH")) cur = std::min<int64_t>(FB, std::max(1, atoi(be)));   // experiments
            void *d_Q, *d_flag;
            AMOF_TRY(ensure(ctx, SLOT_AUX1, (size_t)FB * t->n_atoms * sizeof(QAtom), &d_Q));
            AMOF_TRY(ensure(ctx, SLOT_FLAGS, sizeof(int32_t), &d_flag));
            AMOF_HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, sizeof(int32_t), ctx->stream));
            fa.Q = (const QAtom *)d_Q;
            size_t lds = 2 * (FAST_TILE + FAST_SUB) * sizeof(uint4) + (size_t)((nbins + FAST_TRASH + 2 + 3) & ~3) * sizeof(unsigned);
            fa.img_queue = 0;
            fa.img_defer = 0;
            const double *d_fold = nullptr;
            RdfFastArgs ft = fa;        // TRI: its own per-cell records, guards and queue
            if (tri) {
                std::vector<FrameScale> fst((size_t)nc);
                double hb = 0.0, gfrac = 0.0;
                for (int64_t k = 0; k < nc; k++) {
                    FrameScale &r = fst[(size_t)k];
                    memset(&r, 0, sizeof r);
                    const double *tr = &tri_rec[(size_t)k * 9];
                    r.sc64[0] = tr[0]; r.sc64[3] = tr[1]; r.sc64[4] = tr[2]; r.sc64[8] = tr[3];
                    r.sc[3] = (float)(tr[0] * tr[0]); r.sc[4] = (float)(tr[2] * tr[2]); r.sc[5] = (float)(tr[3] * tr[3]);
                    r.sc[8] = (float)tr[3];
                    r.tri_c10 = (float)(tr[1] / tr[0]);
                    auto down = [](double v) {      // largest float <= v
                        if (!(v < INFINITY)) return (float)INFINITY;
                        float f = (float)v;
                        if ((double)f > v) f = nextafterf(f, -INFINITY);
                        return f;
                    };
                    r.tri_near_y = down(tr[4]);
                    r.tri_near_z = down(tr[5]);
                    r.tri_near_x = down(tr[8]);
                    r.tri_kx = (uint32_t)(long long)rint(tr[6] * 4294967296.0);
                    r.tri_ky = (uint32_t)(long long)rint(tr[7] * 4294967296.0);
                    r.cull_gap = fsv[(size_t)k].cull_gap;           // (same slab axis, same height)
                    for (int q = 0; q < 3; q++) r.near_t[q] = 0x7fffffffu;
                    hb = std::max(hb, geom.rec[(size_t)k * GEOM_STRIDE + 18 + axis] / dr);
                    gfrac = std::max(gfrac, cull ? (double)r.cull_gap / 4294967296.0 : 0.5);
                }
                void *d_fst, *d_foldv;
                AMOF_TRY(upload(ctx, SLOT_AUX5, fst.data(), fst.size() * sizeof(FrameScale), &d_fst));
                AMOF_TRY(upload(ctx, SLOT_AUX6, tri_fold.data(), tri_fold.size() * sizeof(double), &d_foldv));
                d_fold = (const double *)d_foldv;
                ft.fs = (const FrameScale *)d_fst;
                // folded coordinates: two fold roundings and the wrap constant on top of the truncation (2.5 grid units per
                // axis instead of 1): twice the grid term; XW: the truncated f32 product c10 iy moves the x wrap and the
                // candidate by < 1 + 256 |c10| units more (quant = 2 units of csum)
                // (in units of the x axis: |A| <= csum; 1.5 for margin)
                const double guard_m_tri = 2.0 * quant / dr + (tri_code >= 5 || tri_code % 5 == 4 ? (1.0 + 256.0 * tri_c10) * 1.5 * csum * two32_ / dr : 0.0) +
                                           (double)nbins * 1e-12;
                // (the twin image of near mode 4 has |iy| up to 2^31 (1 + 2 tau): its candidate's error bound grows with it)
                const double guard_tri = fast_guard_tri(nbins, hb, gfrac, tri_l10_bins) * (1.0 + 4.0 * tri_tau) + guard_m_tri;
                if (!(guard_tri < 0.25)) return fail(ctx, AMOF_ECAPACITY, "TRI guard out of range");      // (checked at selection)
                const double half = 0.5 - guard_tri;
                float fhalf = (float)half;
                if ((double)fhalf > half) fhalf = nextafterf(fhalf, -INFINITY);
                ft.half_m_guard = fhalf;
                ft.guard64 = guard_m_tri;
                ft.guard64_2 = 2.0 * guard_m_tri * (1.0 + 1e-9);
                ft.guard64_sq = guard_m_tri * guard_m_tri * (1.0 + 1e-9);
                // queue: the near pairs (share of all pairs) and the level-3 pairs of a step
                const double expect = tri_share * (double)FAST_SUB * FAST_TILE;
                const double want = std::max(96.0, ceil(4.0 * expect / 32.0) * 32.0);
                ft.img_defer = want <= 256.0 ? 1 : 0;
                ft.img_queue = ft.img_defer ? (int32_t)want : IMG_QUEUE_MAX;
                lds = 2 * (FAST_TILE + FAST_SUB) * sizeof(uint4) + (size_t)((nbins + FAST_TRASH + 2 + 3) & ~3) * sizeof(unsigned) +
                      (ft.img_defer ? 2 : 1) * (size_t)ft.img_queue * sizeof(uint2) + 16;
            }
            if (fast_img) {
                // expected number of parked pairs per step (a step is 128 x 512 pairs).  Slightly sheared cells: two
                // buffers of >= 96 entries, evaluated a step later -- small enough for five workgroups per CU and no
                // extra barrier; larger shares: one buffer of 1024, evaluated at the end of the step.
                const double expect = img_share * (double)FAST_SUB * FAST_TILE;
                const double want = std::max(96.0, ceil(4.0 * expect / 32.0) * 32.0);
                fa.img_defer = want <= 256.0 ? 1 : 0;
                fa.img_queue = fa.img_defer ? (int32_t)want : IMG_QUEUE_MAX;
                lds = 2 * (FAST_TILE + FAST_SUB) * sizeof(uint4) + (size_t)((nbins + FAST_TRASH + 2 + 3) & ~3) * sizeof(unsigned) +
                      (fa.img_defer ? 2 : 1) * (size_t)fa.img_queue * sizeof(uint2) + 16;
            }
            int64_t launches = 0;
            for (int64_t fb = 0; fb < t->n_frames; fb += cur, cur = std::min<int64_t>(2 * cur, FB)) {
                const int64_t nf = std::min<int64_t>(cur, t->n_frames - fb);
                AMOF_TRY(stager_need(stage, fb + nf));
                AMOF_TRY(launch_quantize(ctx, pos_dev, (const double *)d_geom, (int)t->n_cells, (const int32_t *)d_perm,
                                         (const int64_t *)d_spfirst, S, t->n_atoms, (int)fb, (int)nf, axis, (QAtom *)d_Q,
                                         nullptr, (int32_t *)d_flag, tri ? tri_ax0 : -1, tri ? tri_ax1 : -1, d_fold));
                fa.f_base = (int32_t)fb;
                fa.nf = (int32_t)nf;
                // frames per workgroup chunk: ~80k workgroups per launch (1280 run at a time: > 60 rounds, so that
                // ramp-up and drain stay around 1-2 %), between 2 and 16 frames -- fewer frames flush the LDS histogram
                // more often (u64 global atomics: HBM-side write traffic), more do not fit the 4 MiB L2 of an XCD.
                // Measured on cfg3 (profiles/r02/rdf_fpc_sweep.txt): a 625-frame shard 11.33 (10 frames) -> 11.08 ms (2);
                // 5000 frames stay at 16 (8 would be 0.5 % faster for 38 % more traffic).
                int64_t fpc = (nf * (int64_t)fpairs.size() + 40000) / 80000;
                fpc = std::max<int64_t>(2, std::min<int64_t>(fpc, 16));
                if (const char *fpc_env = getenv("AMOF_RDF_FPC")) fpc = std::max(1, atoi(fpc_env));     // experiments
                fpc = std::min<int64_t>(fpc, std::max<int64_t>(1, nf));
                if (nf >= 64) fpc = std::min<int64_t>(fpc, nf / 32);   // >= 32 chunks: every XCD gets >= 4
                int64_t chunks = (nf + fpc - 1) / fpc;
                fa.xcd_map = chunks >= 32 ? 1 : 0;
                // the XCD mapping deals chunks in groups of 8, over a number of frames that divides by 8; the rest: one grid
                // row per frame behind the chunks (rdf_tile_kernel_fast)
                const int64_t nf_tail = fa.xcd_map && !getenv("AMOF_RDF_NOTAIL") ? nf % 8 : 0;
                if (fa.xcd_map) chunks = ((nf - nf_tail + fpc - 1) / fpc + 7) / 8 * 8;
                fa.a.frames_per_chunk = (int32_t)fpc;
                fa.n_chunks = (int32_t)chunks;
                fa.nf_main = (int32_t)(nf - nf_tail);
                dim3 grid((unsigned)fpairs.size(), (unsigned)(chunks + nf_tail));
                if (launches == 0) timing_dom_begin(ctx, tri ? "rdf_tile_tri" : fast_img ? "rdf_tile_img" : use_zf ? "rdf_tile_zf" : "rdf_tile");
                auto launch = [&](auto kern) -> hipError_t {
                    hipError_t e2 = allow_max_lds_from_zero((const void *)kern);
                    if (e2 != hipSuccess) return e2;
                    hipLaunchKernelGGL(kern, grid, dim3(FAST_THREADS), lds, ctx->stream, fa);
                    return hipSuccess;
                };
                hipError_t e;
                if (tri) {
                    ft.Q = fa.Q; ft.f_base = fa.f_base; ft.nf = fa.nf; ft.xcd_map = fa.xcd_map; ft.n_chunks = fa.n_chunks; ft.nf_main = fa.nf_main;
                    ft.a.frames_per_chunk = fa.a.frames_per_chunk;
                    auto launch_tri = [&](auto kern) -> hipError_t {
                        hipError_t e2 = allow_max_lds_from_zero((const void *)kern);
                        if (e2 != hipSuccess) return e2;
                        hipLaunchKernelGGL(kern, grid, dim3(FAST_THREADS), lds, ctx->stream, ft);
                        return hipSuccess;
                    };
                    // (one instantiation per code: near tests x x wrap)
#define AMOF_TRI(C) (cull ? launch_tri(rdf_tile_kernel_fast<false, true, false, true, C>) : launch_tri(rdf_tile_kernel_fast<false, false, false, true, C>))
                    switch (tri_code) {
                    case 0: e = AMOF_TRI(0); break;
                    case 1: e = AMOF_TRI(1); break;
                    case 2: e = AMOF_TRI(2); break;
                    case 3: e = AMOF_TRI(3); break;
                    case 4: e = AMOF_TRI(4); break;
                    case 5: e = AMOF_TRI(5); break;
                    case 6: e = AMOF_TRI(6); break;
                    case 7: e = AMOF_TRI(7); break;
                    case 8: e = AMOF_TRI(8); break;
                    case 9: e = AMOF_TRI(9); break;
                    case 10: e = AMOF_TRI(10); break;
                    default: e = AMOF_TRI(11); break;
                    }
#undef AMOF_TRI
                }
                else if (fast_img) {
                    if (ortho && cull) e = launch(rdf_tile_kernel_fast<true, true, true>);
                    else if (ortho) e = launch(rdf_tile_kernel_fast<true, false, true>);
                    else if (cull) e = launch(rdf_tile_kernel_fast<false, true, true>);
                    else e = launch(rdf_tile_kernel_fast<false, false, true>);
                }
                else if (use_zf) {
                    RdfFastArgs fz = fa;
                    fz.nb_hi = zf_nb_hi;
                    fz.half_m_guard = zf_half_m_guard;
                    hipError_t e2 = cull ? allow_max_lds_from_zero((const void *)rdf_tile_kernel_fast<true, true, false, true>)
                                         : allow_max_lds_from_zero((const void *)rdf_tile_kernel_fast<true, false, false, true>);
                    if (e2 == hipSuccess && cull)
                        hipLaunchKernelGGL((rdf_tile_kernel_fast<true, true, false, true>), grid, dim3(FAST_THREADS), lds,
                                           ctx->stream, fz);
                    else if (e2 == hipSuccess)
                        hipLaunchKernelGGL((rdf_tile_kernel_fast<true, false, false, true>), grid, dim3(FAST_THREADS), lds,
                                           ctx->stream, fz);
                    e = e2;
                }
                else if (ortho && cull) e = launch(rdf_tile_kernel_fast<true, true>);
                else if (ortho) e = launch(rdf_tile_kernel_fast<true, false>);
                else if (cull) e = launch(rdf_tile_kernel_fast<false, true>);
                else e = launch(rdf_tile_kernel_fast<false, false>);
                AMOF_HIP_TRY(ctx, e);
                AMOF_HIP_TRY(ctx, hipGetLastError());
                launches++;
            }
            timing_dom_end(ctx, launches);
            int32_t flag = 0;
            AMOF_TRY(fetch(ctx, &flag, d_flag, sizeof flag));
            AMOF_HIP_TRY(ctx, sync_stream(ctx));
            if (flag) {
                // some atom lies > 1e4 cells away from the origin: redo with the exact kernel
                AMOF_HIP_TRY(ctx, hipMemsetAsync(d_U, 0, U_bytes, ctx->stream));
            } else {
                done = true;
            }
            }   // tile kernel (not the range kernel)
        }
        if (!done) {
        AMOF_TRY(stager_need(stage, t->n_frames));
        unsigned chunks;
        pick_chunks(t->n_frames, a.frames_per_chunk, chunks);
        dim3 grid((unsigned)pairs.size(), chunks);
        timing_dom_begin(ctx, "rdf_exact");
        if (nbins <= AMOF_MAX_LDS_BINS) {
            size_t lds = 3 * RDF_TILE * sizeof(double) + (size_t)nbins * sizeof(unsigned);
            auto launch = [&](auto kern) -> hipError_t {
                hipError_t e = allow_max_lds((const void *)kern);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, grid, dim3(RDF_TILE), lds, ctx->stream, a);
                return hipGetLastError();
            };
            hipError_t e;
            if (ortho && !extra) e = launch(rdf_tile_kernel<true, false>);
            else if (ortho && extra) e = launch(rdf_tile_kernel<true, true>);
            else if (!ortho && !extra) e = launch(rdf_tile_kernel<false, false>);
            else e = launch(rdf_tile_kernel<false, true>);
            AMOF_HIP_TRY(ctx, e);
        } else {
            if (ortho && !extra) hipLaunchKernelGGL((rdf_tile_kernel_global<true, false>), grid, dim3(RDF_TILE), 0, ctx->stream, a);
            else if (ortho && extra) hipLaunchKernelGGL((rdf_tile_kernel_global<true, true>), grid, dim3(RDF_TILE), 0, ctx->stream, a);
            else if (!ortho && !extra) hipLaunchKernelGGL((rdf_tile_kernel_global<false, false>), grid, dim3(RDF_TILE), 0, ctx->stream, a);
            else hipLaunchKernelGGL((rdf_tile_kernel_global<false, true>), grid, dim3(RDF_TILE), 0, ctx->stream, a);
            AMOF_HIP_TRY(ctx, hipGetLastError());
        }
        timing_dom_end(ctx, 1);
        }
    }
    {
        size_t total = (size_t)S * S * nbins;
        int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(rdf_finalize_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                           (const unsigned long long *)d_U, (const unsigned long long *)d_self,
                           (const long long *)d_nsp, hist_dev, S, nbins);
        AMOF_HIP_TRY(ctx, hipGetLastError());
    }
    timing_end(ctx);
    // host metadata above lives on this stack frame: finish before returning
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    if (volume_sum) *volume_sum += geom.volume_sum;
    return AMOF_OK;
}

static int rdf_check(amof_ctx *ctx, const amof_traj *t, double rmax, int32_t nbins, const void *hist)
{
    AMOF_TRY(validate_traj(ctx, t, false));
    if (!(rmax > 0.0) || !isfinite(rmax)) return fail(ctx, AMOF_EINVAL, "rmax must be positive and finite");
    if (nbins <= 0) return fail(ctx, AMOF_EINVAL, "nbins must be positive");
    if (!hist) return fail(ctx, AMOF_EINVAL, "hist is NULL");
    return AMOF_OK;
}

}  // namespace amof

using namespace amof;

extern "C" int amof_rdf_accumulate_dev(amof_ctx *ctx, const amof_traj *traj, double rmax, int32_t nbins,
                                       uint64_t *hist_dev, double *volume_sum)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(rdf_check(ctx, traj, rmax, nbins, hist_dev));
    return rdf_run(ctx, traj, rmax, nbins, (unsigned long long *)hist_dev, volume_sum);
}

extern "C" int amof_rdf_accumulate(amof_ctx *ctx, const amof_traj *traj, double rmax, int32_t nbins,
                                   uint64_t *hist, double *volume_sum)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(rdf_check(ctx, traj, rmax, nbins, hist));
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t bytes = (size_t)traj->n_species * traj->n_species * nbins * sizeof(uint64_t);
    void *d_hist = nullptr;
    AMOF_TRY(upload(ctx, SLOT_OUT0, hist, bytes, &d_hist));
    AMOF_TRY(rdf_run(ctx, traj, rmax, nbins, (unsigned long long *)d_hist, volume_sum));
    AMOF_TRY(fetch(ctx, hist, d_hist, bytes));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}
