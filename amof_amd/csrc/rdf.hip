// RDF pair-distance histogram kernels (gfx950).
//
// Replaces asap3's RawRDF as driven by the reference at amof/rdf.py:88-93.
//
// Work decomposition: atoms are sorted by species once per call (the species
// of an atom never change along a trajectory) and cut into species-pure tiles
// of <= 256 atoms.  A workgroup owns ONE unordered tile pair (I <= J) and a
// chunk of consecutive frames: every thread keeps one atom of tile I in
// registers, tile J is staged in LDS and read by broadcast, and the pair's
// species key is uniform per workgroup, so a single nbins-wide u32 histogram in
// LDS serves the whole workgroup.  It is flushed with u64 global atomics once
// per frame chunk.  Nothing but the 24*N bytes of a frame is read from HBM per
// frame; every frame is re-read ~2*ntiles times from L2, never from HBM.
#include <math.h>

#include <algorithm>
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
    const double *pos_dev = nullptr;
    AMOF_TRY(stage_positions(ctx, t, &pos_dev));
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
        // enough workgroups to fill 256 CUs several times over, few enough
        // flushes that the u64 global atomics stay negligible
        int64_t want_chunks = (8 * 2048 + (int64_t)pairs.size() - 1) / (int64_t)pairs.size();
        int64_t fpc = std::max<int64_t>(1, t->n_frames / std::max<int64_t>(1, want_chunks));
        fpc = std::min<int64_t>(fpc, 64);
        int64_t chunks = (t->n_frames + fpc - 1) / fpc;
        if (chunks > 65535) {
            fpc = (t->n_frames + 65534) / 65535;
            chunks = (t->n_frames + fpc - 1) / fpc;
        }
        a.frames_per_chunk = (int32_t)fpc;
        dim3 grid((unsigned)pairs.size(), (unsigned)chunks);
        const bool extra = max_img > 0;
        const bool ortho = geom.all_ortho;
        timing_dom_begin(ctx);
        if (nbins <= AMOF_MAX_LDS_BINS) {
            size_t lds = 3 * RDF_TILE * sizeof(double) + (size_t)nbins * sizeof(unsigned);
            auto launch = [&](auto kern) -> hipError_t {
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds);
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
    AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
    AMOF_HIP_TRY(ctx, hipMemcpyAsync(hist, d_hist, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMOF_OK;
}
