/*
 * amof_oracle.c -- CPU restatement of aMOF's per-frame pair-distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed
 * "port" CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (amof_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for RDF / CN / BAD.  The reference
 * (coudertlab/amof v1.1.0) holds no tests or golden vectors (amof/tests/
 * __init__.py:1-4) and delegates the arithmetic of these three analyses to
 * un-vendored third-party packages that are absent here:
 *   - asap3==3.12.8  (requirements.txt:1)  RadialDistributionFunction
 *       call sites amof/rdf.py:90-96,109
 *   - ase==3.20.1    (requirements.txt:2)  neighbor_list / get_angles / find_mic
 *       call sites amof/atom.py:82, amof/bad.py:100
 * Their published behaviour is restated below; what pins it are analytic
 * known-answer tests, sum rules and the reference fixture's known answers
 * (tests/test_oracle_*.py).  The MSD restatement (oracle/numpy_oracle.py) IS
 * pinned by vectors generated from the reference's own compute_msd_of_m.
 *
 * Canonical pair arithmetic (shared definition with the HIP kernels so that
 * integer results are bit-identical):
 *   d0 = r_j - r_i                       (3 IEEE subtractions, raw positions)
 *   s_k = fma(d0z, inv[2][k], fma(d0y, inv[1][k], d0x*inv[0][k]))
 *   n_k = rint(s_k)                      (ties to even; 0 on non-periodic axes
 *                                         because that inverse column is zeroed)
 *   d_c = fma(-n2, C[2][c], fma(-n1, C[1][c], fma(-n0, C[0][c], d0_c)))
 *   d2  = fma(dz, dz, fma(dy, dy, dx*dx))
 * plus, for small or skewed cells, every further periodic image d + E_m for
 * the lattice vectors E_m that can reach inside the cutoff ("image complete":
 * asap3 and ASE both count true periodic images).
 * RDF bin: b = (int)(sqrt(d2) / (rmax/nbins)), counted iff d2 < rmax^2 and
 * b < nbins (asap3 RawRDF semantics, [3P-memory], rdf.py:90-93).
 * CN / BAD neighbour: sqrt(d2) < rc  (strict, ASE neighbor_list, atom.py:82).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AMOF_OK 0
#define AMOF_EINVAL (-1)
#define AMOF_ESINGULAR (-2)
#define AMOF_EANGLE (-3)
#define AMOF_ENOMEM (-4)

#define MAX_IMG 4096

typedef struct {
    double c[9];    /* cell rows */
    double inv[9];  /* cell^-1 with non-periodic columns zeroed: s = d . inv */
    double invf[9]; /* full inverse */
    double h[3];    /* perpendicular heights */
    double vol;
} geom_t;

/* reference: ase cell conventions (row vectors), amof/atom.py:22 get_volume */
static int geom_make(const double *c, const unsigned char *pbc, geom_t *g)
{
    double m00 = c[4] * c[8] - c[5] * c[7];
    double m01 = c[3] * c[8] - c[5] * c[6];
    double m02 = c[3] * c[7] - c[4] * c[6];
    double det = c[0] * m00 - c[1] * m01 + c[2] * m02;
    if (!(fabs(det) > 0.0) || !isfinite(det)) return AMOF_ESINGULAR;
    memcpy(g->c, c, sizeof(double) * 9);
    g->invf[0] = (c[4] * c[8] - c[5] * c[7]) / det;
    g->invf[1] = (c[2] * c[7] - c[1] * c[8]) / det;
    g->invf[2] = (c[1] * c[5] - c[2] * c[4]) / det;
    g->invf[3] = (c[5] * c[6] - c[3] * c[8]) / det;
    g->invf[4] = (c[0] * c[8] - c[2] * c[6]) / det;
    g->invf[5] = (c[2] * c[3] - c[0] * c[5]) / det;
    g->invf[6] = (c[3] * c[7] - c[4] * c[6]) / det;
    g->invf[7] = (c[1] * c[6] - c[0] * c[7]) / det;
    g->invf[8] = (c[0] * c[4] - c[1] * c[3]) / det;
    for (int k = 0; k < 3; k++) {
        double cx = g->invf[k], cy = g->invf[3 + k], cz = g->invf[6 + k];
        g->h[k] = 1.0 / sqrt(cx * cx + cy * cy + cz * cz);
        for (int i = 0; i < 3; i++) g->inv[3 * i + k] = pbc[k] ? g->invf[3 * i + k] : 0.0;
    }
    g->vol = fabs(det);
    return AMOF_OK;
}

/* Lattice vectors E != 0 that can bring a base-image vector (fractional
 * components in [-1/2,1/2]) within R: keep n iff
 *   max_k h_k * max(0, |n_k| - 1/2) < R * (1 - 1e-9).
 * The relative margin makes the default rmax = half the shortest cell length
 * (rdf.py:74) yield an empty list for orthogonal cells. */
static int images_make(const geom_t *g, const unsigned char *pbc, double R, double *E, int *n_img)
{
    int M[3];
    double Rm = R * (1.0 - 1e-9);
    for (int k = 0; k < 3; k++) {
        if (!pbc[k]) { M[k] = 0; continue; }
        double q = floor(Rm / g->h[k] + 0.5) + 1.0;
        if (q > 64) return AMOF_EINVAL;
        M[k] = (int)q;
    }
    int cnt = 0;
    for (int a = -M[0]; a <= M[0]; a++)
        for (int b = -M[1]; b <= M[1]; b++)
            for (int cc = -M[2]; cc <= M[2]; cc++) {
                if (a == 0 && b == 0 && cc == 0) continue;
                int n[3] = {a, b, cc};
                double bound = 0.0;
                for (int k = 0; k < 3; k++) {
                    double t = fabs((double)n[k]) - 0.5;
                    if (t > 0.0 && g->h[k] * t > bound) bound = g->h[k] * t;
                }
                if (!(bound < Rm)) continue;
                if (cnt >= MAX_IMG) return AMOF_EINVAL;
                for (int c = 0; c < 3; c++)
                    E[3 * cnt + c] = fma((double)n[2], g->c[6 + c],
                                         fma((double)n[1], g->c[3 + c], (double)n[0] * g->c[c]));
                cnt++;
            }
    *n_img = cnt;
    return AMOF_OK;
}

static inline void pair_base(const geom_t *g, double d0x, double d0y, double d0z,
                             double *dx, double *dy, double *dz)
{
    const double *inv = g->inv, *c = g->c;
    double s0 = fma(d0z, inv[6], fma(d0y, inv[3], d0x * inv[0]));
    double s1 = fma(d0z, inv[7], fma(d0y, inv[4], d0x * inv[1]));
    double s2 = fma(d0z, inv[8], fma(d0y, inv[5], d0x * inv[2]));
    double n0 = rint(s0), n1 = rint(s1), n2 = rint(s2);
    *dx = fma(-n2, c[6], fma(-n1, c[3], fma(-n0, c[0], d0x)));
    *dy = fma(-n2, c[7], fma(-n1, c[4], fma(-n0, c[1], d0y)));
    *dz = fma(-n2, c[8], fma(-n1, c[5], fma(-n0, c[2], d0z)));
}

static inline double norm2(double dx, double dy, double dz)
{
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

/* exported for tests: geometry record = cell[9], inv[9], h[3], vol  (22 doubles) */
int amof_oracle_geom(const double *cell, const unsigned char *pbc, double *out22)
{
    geom_t g;
    int rc = geom_make(cell, pbc, &g);
    if (rc) return rc;
    memcpy(out22, g.c, 72);
    memcpy(out22 + 9, g.inv, 72);
    memcpy(out22 + 18, g.h, 24);
    out22[21] = g.vol;
    return AMOF_OK;
}

int amof_oracle_images(const double *cell, const unsigned char *pbc, double R,
                       double *E, int max_img, int *n_img)
{
    geom_t g;
    static double tmp[3 * MAX_IMG];
    int rc = geom_make(cell, pbc, &g);
    if (rc) return rc;
    int n = 0;
    rc = images_make(&g, pbc, R, tmp, &n);
    if (rc) return rc;
    if (n > max_img) return AMOF_EINVAL;
    memcpy(E, tmp, sizeof(double) * 3 * n);
    *n_img = n;
    return AMOF_OK;
}

/* ------------------------------------------------------------------ RDF -- */

typedef struct {
    double rmax2, dr;
    int nbins, S;
    const int *species;
    uint64_t *U; /* [S*S][nbins] unordered-pair counts, key (min,max) */
} rdf_acc_t;

static inline void rdf_count(rdf_acc_t *a, int si, int sj, double d2)
{
    if (d2 < a->rmax2) {
        int b = (int)(sqrt(d2) / a->dr);
        if (b < a->nbins) {
            int lo = si < sj ? si : sj, hi = si < sj ? sj : si;
            a->U[((size_t)lo * a->S + hi) * a->nbins + b]++;
        }
    }
}

static inline void rdf_pair(rdf_acc_t *a, const geom_t *g, const double *E, int nE,
                            const double *p, int i, int j)
{
    double dx, dy, dz;
    pair_base(g, p[3 * j] - p[3 * i], p[3 * j + 1] - p[3 * i + 1], p[3 * j + 2] - p[3 * i + 2],
              &dx, &dy, &dz);
    int si = a->species[i], sj = a->species[j];
    rdf_count(a, si, sj, norm2(dx, dy, dz));
    for (int m = 0; m < nE; m++)
        rdf_count(a, si, sj, norm2(dx + E[3 * m], dy + E[3 * m + 1], dz + E[3 * m + 2]));
}

/* candidate pruning grid (cell list).  Only selects which (i<j) pairs are
 * evaluated; every evaluated pair uses the canonical arithmetic above, so the
 * counts equal the brute-force ones. */
typedef struct {
    int g[3];
    int *cell_of, *start, *order;
} grid_t;

static int grid_build(grid_t *G, const geom_t *gm, const unsigned char *pbc,
                      const double *p, int64_t N, double R)
{
    double cs = R / 3.0;
    double dens = cbrt(gm->vol / (double)N * 4.0);
    if (cs < dens) cs = dens;
    int nc = 1;
    for (int k = 0; k < 3; k++) {
        int gk = pbc[k] ? (int)floor(gm->h[k] / cs) : 1;
        if (gk < 1) gk = 1;
        if (gk > 128) gk = 128;
        G->g[k] = gk;
        nc *= gk;
    }
    G->cell_of = (int *)malloc(sizeof(int) * N);
    G->order = (int *)malloc(sizeof(int) * N);
    G->start = (int *)calloc(nc + 1, sizeof(int));
    if (!G->cell_of || !G->order || !G->start) return AMOF_ENOMEM;
    for (int64_t i = 0; i < N; i++) {
        int idx[3];
        for (int k = 0; k < 3; k++) {
            double s = p[3 * i] * gm->invf[k] + p[3 * i + 1] * gm->invf[3 + k] + p[3 * i + 2] * gm->invf[6 + k];
            s -= floor(s);
            int q = (int)(s * G->g[k]);
            if (q >= G->g[k]) q = G->g[k] - 1;
            if (q < 0) q = 0;
            idx[k] = q;
        }
        int cidx = (idx[0] * G->g[1] + idx[1]) * G->g[2] + idx[2];
        G->cell_of[i] = cidx;
        G->start[cidx + 1]++;
    }
    for (int cix = 0; cix < nc; cix++) G->start[cix + 1] += G->start[cix];
    int *fill = (int *)calloc(nc, sizeof(int));
    if (!fill) return AMOF_ENOMEM;
    for (int64_t i = 0; i < N; i++) {
        int cidx = G->cell_of[i];
        G->order[G->start[cidx] + fill[cidx]++] = (int)i;
    }
    free(fill);
    return AMOF_OK;
}

static void grid_free(grid_t *G)
{
    free(G->cell_of); free(G->start); free(G->order);
}

/* distinct cells along axis k reachable from index q within R */
static int reach_list(const grid_t *G, const geom_t *gm, int k, int q, double R, int *out)
{
    int gk = G->g[k];
    double w = gm->h[k] / gk;
    int m = (int)floor(R * (1.0 + 1e-9) / w) + 1;
    if (2 * m + 1 >= gk) {
        for (int t = 0; t < gk; t++) out[t] = t;
        return gk;
    }
    int n = 0;
    for (int t = -m; t <= m; t++) out[n++] = ((q + t) % gk + gk) % gk;
    return n;
}

/*
 * Accumulate the S*S ordered-pair RDF histograms of a trajectory.
 * Restates asap3 RadialDistributionFunction.update() as consumed by
 * amof/rdf.py:88-93 ([3P-memory]; see header).
 *   hist[(a*S+b)*nbins + bin] += #ordered pairs (i of species a, j of species b)
 *   volume_sum += sum of frame volumes (asap3 accumulates it for normalisation)
 */
int amof_oracle_rdf(const double *pos, const double *cell, int64_t n_cells,
                    const unsigned char *pbc, int64_t F, int64_t N, const int *species, int S,
                    double rmax, int nbins, uint64_t *hist, double *volume_sum, int use_cell_list)
{
    if (F < 0 || N < 0 || S <= 0 || nbins <= 0 || !(rmax > 0.0)) return AMOF_EINVAL;
    if (n_cells != 1 && n_cells != F) return AMOF_EINVAL;
    for (int64_t i = 0; i < N; i++) if (species[i] < 0 || species[i] >= S) return AMOF_EINVAL;
    rdf_acc_t acc;
    acc.rmax2 = rmax * rmax;
    acc.dr = rmax / nbins;
    acc.nbins = nbins;
    acc.S = S;
    acc.species = species;
    acc.U = (uint64_t *)calloc((size_t)S * S * nbins, sizeof(uint64_t));
    uint64_t *selfh = (uint64_t *)calloc((size_t)nbins, sizeof(uint64_t));
    int64_t *nsp = (int64_t *)calloc(S, sizeof(int64_t));
    double *E = (double *)malloc(sizeof(double) * 3 * MAX_IMG);
    if (!acc.U || !selfh || !nsp || !E) return AMOF_ENOMEM;
    for (int64_t i = 0; i < N; i++) nsp[species[i]]++;
    double vsum = 0.0;
    int rc = AMOF_OK;
    for (int64_t f = 0; f < F && rc == AMOF_OK; f++) {
        geom_t g;
        int nE = 0;
        rc = geom_make(cell + 9 * (n_cells == 1 ? 0 : f), pbc, &g);
        if (rc) break;
        rc = images_make(&g, pbc, rmax, E, &nE);
        if (rc) break;
        vsum += g.vol;
        const double *p = pos + (size_t)f * N * 3;
        /* self images (i,i,E): position independent */
        for (int m = 0; m < nE; m++) {
            double d2 = norm2(E[3 * m], E[3 * m + 1], E[3 * m + 2]);
            if (d2 < acc.rmax2) {
                int b = (int)(sqrt(d2) / acc.dr);
                if (b < nbins) selfh[b]++;
            }
        }
        if (!use_cell_list) {
            for (int64_t i = 0; i < N; i++)
                for (int64_t j = i + 1; j < N; j++)
                    rdf_pair(&acc, &g, E, nE, p, (int)i, (int)j);
        } else {
            grid_t G;
            rc = grid_build(&G, &g, pbc, p, N, rmax);
            if (rc) break;
            int lx[260], ly[260], lz[260];
            for (int64_t i = 0; i < N; i++) {
                int ci = G.cell_of[i];
                int qz = ci % G.g[2], qy = (ci / G.g[2]) % G.g[1], qx = ci / (G.g[2] * G.g[1]);
                int nx = reach_list(&G, &g, 0, qx, rmax, lx);
                int ny = reach_list(&G, &g, 1, qy, rmax, ly);
                int nz = reach_list(&G, &g, 2, qz, rmax, lz);
                for (int a = 0; a < nx; a++)
                    for (int b = 0; b < ny; b++)
                        for (int c = 0; c < nz; c++) {
                            int cj = (lx[a] * G.g[1] + ly[b]) * G.g[2] + lz[c];
                            for (int t = G.start[cj]; t < G.start[cj + 1]; t++) {
                                int j = G.order[t];
                                if (j > i) rdf_pair(&acc, &g, E, nE, p, (int)i, j);
                            }
                        }
            }
            grid_free(&G);
        }
    }
    if (rc == AMOF_OK) {
        for (int a = 0; a < S; a++)
            for (int b = 0; b < S; b++) {
                int lo = a < b ? a : b, hi = a < b ? b : a;
                const uint64_t *u = acc.U + ((size_t)lo * S + hi) * nbins;
                uint64_t *h = hist + ((size_t)a * S + b) * nbins;
                for (int k = 0; k < nbins; k++)
                    h[k] += (a == b) ? 2 * u[k] + (uint64_t)nsp[a] * selfh[k] : u[k];
            }
        if (volume_sum) *volume_sum += vsum;
    }
    free(acc.U); free(selfh); free(nsp); free(E);
    return rc;
}

/* ------------------------------------------------------- neighbour lists -- */

typedef struct {
    int j;
    double v[3]; /* minimum-image vector r_j - r_i (what ase get_angles(mic=True) uses) */
} nb_t;

typedef struct {
    nb_t *e;
    int n, cap;
} nblist_t;

static int nb_push(nblist_t *l, int j, double x, double y, double z)
{
    if (l->n == l->cap) {
        int nc = l->cap ? 2 * l->cap : 16;
        nb_t *ne = (nb_t *)realloc(l->e, sizeof(nb_t) * nc);
        if (!ne) return AMOF_ENOMEM;
        l->e = ne;
        l->cap = nc;
    }
    l->e[l->n].j = j;
    l->e[l->n].v[0] = x; l->e[l->n].v[1] = y; l->e[l->n].v[2] = z;
    l->n++;
    return AMOF_OK;
}

/* neighbours of atom i under per-species-pair cutoffs rcm[S*S] (0 = never):
 * every periodic image with sqrt(d2) < rc is one entry (ASE neighbor_list 'ij',
 * amof/atom.py:82-86); no zero-shift self pair.  The stored vector is the
 * minimum-image one (shortest among base + extra images, first on ties). */
static int neighbours_of(const geom_t *g, const double *E, int nE, const double *p, int64_t N,
                         const int *species, int S, const double *rcm, int i, nblist_t *l)
{
    l->n = 0;
    int si = species[i];
    for (int64_t j = 0; j < N; j++) {
        double rc = rcm[si * S + species[j]];
        if (!(rc > 0.0)) continue;
        if (j == i && nE == 0) continue;
        double dx, dy, dz;
        pair_base(g, p[3 * j] - p[3 * i], p[3 * j + 1] - p[3 * i + 1], p[3 * j + 2] - p[3 * i + 2],
                  &dx, &dy, &dz);
        double best2 = norm2(dx, dy, dz), bx = dx, by = dy, bz = dz;
        int hits = 0;
        if (j != i && sqrt(best2) < rc) hits++;
        for (int m = 0; m < nE; m++) {
            double ex = dx + E[3 * m], ey = dy + E[3 * m + 1], ez = dz + E[3 * m + 2];
            double e2 = norm2(ex, ey, ez);
            if (sqrt(e2) < rc) hits++;
            if (e2 < best2) { best2 = e2; bx = ex; by = ey; bz = ez; }
        }
        for (int h = 0; h < hits; h++) {
            int rcode = nb_push(l, (int)j, bx, by, bz);
            if (rcode) return rcode;
        }
    }
    return AMOF_OK;
}

static double max_cutoff(const double *rcm, int S)
{
    double r = 0.0;
    for (int k = 0; k < S * S; k++) if (rcm[k] > r) r = rcm[k];
    return r;
}

/*
 * Coordination numbers.  Restates amof/cn.py:58-73: for each set (A,B),
 * sums[f][s] = sum over atoms i of species A of #neighbours of species B.
 * per_atom (optional) [F][n_sets][N]: that count for every atom (-1 where the
 * atom is not of species A).
 */
int amof_oracle_cn(const double *pos, const double *cell, int64_t n_cells, const unsigned char *pbc,
                   int64_t F, int64_t N, const int *species, int S, const double *rcm,
                   const int *sets, int n_sets, int64_t *sums, int32_t *per_atom)
{
    if (F < 0 || N < 0 || S <= 0 || n_sets < 0) return AMOF_EINVAL;
    if (n_cells != 1 && n_cells != F) return AMOF_EINVAL;
    double R = max_cutoff(rcm, S);
    double *E = (double *)malloc(sizeof(double) * 3 * MAX_IMG);
    if (!E) return AMOF_ENOMEM;
    nblist_t l = {0, 0, 0};
    int rc = AMOF_OK;
    for (int64_t f = 0; f < F && rc == AMOF_OK; f++) {
        geom_t g;
        int nE = 0;
        rc = geom_make(cell + 9 * (n_cells == 1 ? 0 : f), pbc, &g);
        if (rc) break;
        if (R > 0.0) rc = images_make(&g, pbc, R, E, &nE);
        if (rc) break;
        const double *p = pos + (size_t)f * N * 3;
        for (int s = 0; s < n_sets; s++) sums[f * n_sets + s] = 0;
        for (int64_t i = 0; i < N && rc == AMOF_OK; i++) {
            rc = neighbours_of(&g, E, nE, p, N, species, S, rcm, (int)i, &l);
            for (int s = 0; s < n_sets; s++) {
                int A = sets[2 * s], B = sets[2 * s + 1];
                int32_t cnt = -1;
                if (species[i] == A) {
                    cnt = 0;
                    for (int t = 0; t < l.n; t++) if (species[l.e[t].j] == B) cnt++;
                    sums[f * n_sets + s] += cnt;
                }
                if (per_atom) per_atom[((size_t)f * n_sets + s) * N + i] = cnt;
            }
        }
    }
    free(l.e); free(E);
    return rc;
}

/* numpy.histogram with explicit edges: bin k holds edges[k] <= x < edges[k+1],
 * last bin right-closed (reference amof/bad.py:160).  Returns -1 outside. */
static int hist_bin(const double *edges, int nb, double x)
{
    if (!(x >= edges[0]) || !(x <= edges[nb])) return -1;
    int lo = 0, hi = nb; /* find largest k with edges[k] <= x */
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (edges[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

/* arccos with a FIXED algorithm.  ase.geometry.get_angles calls numpy.arccos, whose float64 implementation depends on the
 * CPU numpy was built / dispatched for: on the build machine its SIMD routine is not the correctly rounded value for 9.4 %
 * of arguments (checked against mpmath), glibc 2.35's acos for 0.07 %, the GPU's (ocml) for about as many -- so an angle
 * that lies within one unit in the last place of a histogram edge (exact lattices: 45, 60, 135 degrees ...) is binned
 * by whichever library evaluates it.  The oracle and the GPU kernels therefore both evaluate the published fdlibm
 * algorithm (Sun Microsystems' e_acos.c: rational approximation R(z) = z P(z) / Q(z) of (asin(x) - x) / x with x^2 = z,
 * error < 1 ulp; 0.89 ulp measured), written independently on either side from that description: + - * / sqrt only, every
 * one correctly rounded on both machines, no contraction -- the same bits by construction, and within one unit in the
 * last place of whatever numpy returns. */
static double acos_rational(double z)
{
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
                     z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 +
                     z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}

double amof_oracle_acos(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double ax = fabs(x);
    if (ax >= 1.0) return x > 0.0 ? 0.0 : pi + 2.0 * pio2_lo;          /* (arguments are clipped to [-1, 1] by the caller) */
    if (ax < 0.5) {
        if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;      /* |x| < 2^-57 */
        const double r = acos_rational(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0.0) {                       /* acos(x) = pi - 2 asin(sqrt((1 + x) / 2)) */
        const double z = (1.0 + x) * 0.5, s = sqrt(z);
        const double w = acos_rational(z) * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    {                                    /* acos(x) = 2 asin(sqrt((1 - x) / 2)), the square root split into head + correction */
        const double z = (1.0 - x) * 0.5, s = sqrt(z);
        uint64_t bits;
        double df;
        memcpy(&bits, &s, sizeof bits);
        bits &= 0xffffffff00000000ull;
        memcpy(&df, &bits, sizeof df);
        const double c = (z - df * df) / (s + df);
        const double w = acos_rational(z) * s + c;
        return 2.0 * (df + w);
    }
}

/* ase.geometry.get_angles ([3P-memory], called at amof/bad.py:100): normalise
 * both vectors, dot, clip to [-1,1], arccos (amof_oracle_acos above), degrees = (180/pi) * angle. */
static int angle_deg(const double *v1, const double *v2, double *out)
{
    double n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    double n2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    if (!(n1 > 0.0) || !(n2 > 0.0)) return AMOF_EANGLE;
    double a0 = v1[0] / n1, a1 = v1[1] / n1, a2 = v1[2] / n1;
    double b0 = v2[0] / n2, b1 = v2[1] / n2, b2 = v2[2] / n2;
    double dot = a0 * b0 + a1 * b1 + a2 * b2;
    if (dot > 1.0) dot = 1.0;
    if (dot < -1.0) dot = -1.0;
    *out = (180.0 / M_PI) * amof_oracle_acos(dot);
    return AMOF_OK;
}

/*
 * Bond-angle histograms.  Restates amof/bad.py:70-114,154-160: for each
 * triple t = (A,B) (-1 = "X", any species) and each centre a of species A,
 * every unordered pair of B-neighbours of a contributes the angle B-a-B.
 *   hist[t*nb + k] += count;  n_angles[t] (optional) += number of angles
 *   (angles outside [edges[0], edges[nb]] are dropped, as numpy does)
 */
int amof_oracle_bad(const double *pos, const double *cell, int64_t n_cells, const unsigned char *pbc,
                    int64_t F, int64_t N, const int *species, int S, const double *rcm,
                    const int *triples, int T, const double *edges, int nb,
                    uint64_t *hist, uint64_t *n_angles)
{
    if (F < 0 || N < 0 || S <= 0 || T < 0 || nb <= 0) return AMOF_EINVAL;
    if (n_cells != 1 && n_cells != F) return AMOF_EINVAL;
    double R = max_cutoff(rcm, S);
    double *E = (double *)malloc(sizeof(double) * 3 * MAX_IMG);
    if (!E) return AMOF_ENOMEM;
    nblist_t l = {0, 0, 0};
    int rc = AMOF_OK;
    for (int64_t f = 0; f < F && rc == AMOF_OK; f++) {
        geom_t g;
        int nE = 0;
        rc = geom_make(cell + 9 * (n_cells == 1 ? 0 : f), pbc, &g);
        if (rc) break;
        if (R > 0.0) rc = images_make(&g, pbc, R, E, &nE);
        if (rc) break;
        const double *p = pos + (size_t)f * N * 3;
        for (int64_t a = 0; a < N && rc == AMOF_OK; a++) {
            int need = 0;
            for (int t = 0; t < T; t++) if (triples[2 * t] < 0 || triples[2 * t] == species[a]) need = 1;
            if (!need) continue;
            rc = neighbours_of(&g, E, nE, p, N, species, S, rcm, (int)a, &l);
            if (rc) break;
            for (int t = 0; t < T && rc == AMOF_OK; t++) {
                int A = triples[2 * t], B = triples[2 * t + 1];
                if (!(A < 0 || A == species[a])) continue;
                for (int u = 0; u < l.n && rc == AMOF_OK; u++) {
                    if (!(B < 0 || species[l.e[u].j] == B)) continue;
                    for (int w = u + 1; w < l.n; w++) {
                        if (!(B < 0 || species[l.e[w].j] == B)) continue;
                        double ang;
                        rc = angle_deg(l.e[u].v, l.e[w].v, &ang);
                        if (rc) break;
                        if (n_angles) n_angles[t]++;
                        int k = hist_bin(edges, nb, ang);
                        if (k >= 0) hist[(size_t)t * nb + k]++;
                    }
                }
            }
        }
    }
    free(l.e); free(E);
    return rc;
}

/*
 * Bond-angle histograms keyed by the centre's number of B-neighbours.
 * Restates BadByCn.bad_BAB (amof/bad.py:190-224): slot c of triple t collects the angles of
 * centres that have exactly c neighbours of kind B (c = cn_max also takes larger counts).
 */
int amof_oracle_bad_by_cn(const double *pos, const double *cell, int64_t n_cells, const unsigned char *pbc,
                          int64_t F, int64_t N, const int *species, int S, const double *rcm,
                          const int *triples, int T, const double *edges, int nb, int cn_max,
                          uint64_t *hist, uint64_t *n_angles)
{
    if (F < 0 || N < 0 || S <= 0 || T < 0 || nb <= 0 || cn_max < 1) return AMOF_EINVAL;
    if (n_cells != 1 && n_cells != F) return AMOF_EINVAL;
    double R = max_cutoff(rcm, S);
    double *E = (double *)malloc(sizeof(double) * 3 * MAX_IMG);
    if (!E) return AMOF_ENOMEM;
    nblist_t l = {0, 0, 0};
    int rc = AMOF_OK;
    for (int64_t f = 0; f < F && rc == AMOF_OK; f++) {
        geom_t g;
        int nE = 0;
        rc = geom_make(cell + 9 * (n_cells == 1 ? 0 : f), pbc, &g);
        if (rc) break;
        if (R > 0.0) rc = images_make(&g, pbc, R, E, &nE);
        if (rc) break;
        const double *p = pos + (size_t)f * N * 3;
        for (int64_t a = 0; a < N && rc == AMOF_OK; a++) {
            rc = neighbours_of(&g, E, nE, p, N, species, S, rcm, (int)a, &l);
            if (rc) break;
            for (int t = 0; t < T && rc == AMOF_OK; t++) {
                int A = triples[2 * t], B = triples[2 * t + 1];
                if (!(A < 0 || A == species[a])) continue;
                int cn = 0;
                for (int u = 0; u < l.n; u++) if (B < 0 || species[l.e[u].j] == B) cn++;
                size_t slot = (size_t)t * (cn_max + 1) + (cn < cn_max ? cn : cn_max);
                for (int u = 0; u < l.n && rc == AMOF_OK; u++) {
                    if (!(B < 0 || species[l.e[u].j] == B)) continue;
                    for (int w = u + 1; w < l.n; w++) {
                        if (!(B < 0 || species[l.e[w].j] == B)) continue;
                        double ang;
                        rc = angle_deg(l.e[u].v, l.e[w].v, &ang);
                        if (rc) break;
                        n_angles[slot]++;
                        int k = hist_bin(edges, nb, ang);
                        if (k >= 0) hist[slot * nb + k]++;
                    }
                }
            }
        }
    }
    free(l.e); free(E);
    return rc;
}

/* test hook: all angles of one triple in one frame, in centre-major order */
int amof_oracle_angles(const double *pos, const double *cell, const unsigned char *pbc, int64_t N,
                       const int *species, int S, const double *rcm, int A, int B,
                       double *out, int64_t max_out, int64_t *n_out)
{
    double R = max_cutoff(rcm, S);
    double *E = (double *)malloc(sizeof(double) * 3 * MAX_IMG);
    if (!E) return AMOF_ENOMEM;
    nblist_t l = {0, 0, 0};
    geom_t g;
    int nE = 0;
    int64_t cnt = 0;
    int rc = geom_make(cell, pbc, &g);
    if (!rc && R > 0.0) rc = images_make(&g, pbc, R, E, &nE);
    for (int64_t a = 0; a < N && rc == AMOF_OK; a++) {
        if (!(A < 0 || A == species[a])) continue;
        rc = neighbours_of(&g, E, nE, pos, N, species, S, rcm, (int)a, &l);
        for (int u = 0; u < l.n && rc == AMOF_OK; u++) {
            if (!(B < 0 || species[l.e[u].j] == B)) continue;
            for (int w = u + 1; w < l.n; w++) {
                if (!(B < 0 || species[l.e[w].j] == B)) continue;
                double ang;
                rc = angle_deg(l.e[u].v, l.e[w].v, &ang);
                if (rc) break;
                if (cnt < max_out) out[cnt] = ang;
                cnt++;
            }
        }
    }
    *n_out = cnt;
    free(l.e); free(E);
    return rc;
}

int amof_oracle_has_fma(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("fma") ? 1 : 0;
#else
    return 1;
#endif
}
