// Host-side arithmetic behind the error bounds of the RDF fast paths (no HIP dependency: the CPU test suite compiles
// it with g++, tests/test_guard_math.py).  Included by amof_internal.h.
#pragma once

#include <math.h>

#include <algorithm>

namespace amof {

// General (triclinic) cells on the RDF fast paths: distances are invariant under rotations, so the scale matrix of the
// f32 candidate and of the f64 level-2 refinement is not the cell C (rows in stored axis order) but the lower-triangular
// factor L of its metric, C C^T = L L^T:  |f C| = |f L|, six multiply-adds per pair instead of nine, whatever the cell's
// orientation and the stored axis order.  (Level 3 works on the original positions with the canonical arithmetic.)
// rows: the three lattice vectors in stored order, already scaled; out: L row-major with its zeros.
inline void lower_factor(const double rows[9], double L[9])
{
    double G[3][3];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) G[k][l] = rows[3 * k] * rows[3 * l] + rows[3 * k + 1] * rows[3 * l + 1] + rows[3 * k + 2] * rows[3 * l + 2];
    for (int q = 0; q < 9; q++) L[q] = 0.0;
    L[0] = sqrt(G[0][0]);
    L[3] = G[1][0] / L[0];
    L[4] = sqrt(G[1][1] - L[3] * L[3]);
    L[6] = G[2][0] / L[0];
    L[7] = (G[2][1] - L[6] * L[3]) / L[4];
    L[8] = sqrt(G[2][2] - L[6] * L[6] - L[7] * L[7]);
}
// kappa of the error model above for the chain built on L: P = |L^-1| |L|, sqrt(||P||_1 ||P||_inf)
inline double kappa_lower(const double L[9])
{
    // inverse of a lower-triangular 3x3
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    M[0] = 1.0 / L[0]; M[4] = 1.0 / L[4]; M[8] = 1.0 / L[8];
    M[3] = -L[3] * M[0] * M[4];
    M[7] = -L[7] * M[4] * M[8];
    M[6] = -(L[6] * M[0] + L[7] * M[3]) * M[8];
    double n1 = 0.0, ninf = 0.0, P[3][3];
    for (int r = 0; r < 3; r++)
        for (int x = 0; x < 3; x++) {
            P[r][x] = 0.0;
            for (int m = 0; m < 3; m++) P[r][x] += fabs(M[3 * r + m]) * fabs(L[3 * m + x]);
        }
    for (int r = 0; r < 3; r++) {
        ninf = std::max(ninf, P[r][0] + P[r][1] + P[r][2]);
        n1 = std::max(n1, P[0][r] + P[1][r] + P[2][r]);
    }
    return sqrt(n1 * ninf);
}
// ZF form of the diagonal-cell tile kernel (rdf.hip fast_q_zf): the slab-axis difference is the difference of two f32
// coordinates in bins, Z_j' = fl(c (z_j - z0)), Z_i' = fl(c (z_i - z0)), one rounding each (f64 product -> f32):
//   |dz - Z| <= A + u |Z|,  A = u (|Z_j'| + |Z_i'|) <= u Hb (G / 2^32 + 1/16 + 1/256)
// (Hb = slab-axis height in bins, G the culling reach, sub-tile span < 2^28, slab widening 2^24; 1/128 taken).
// With zeta = Z^2 / T:  |t - T| <= u T (7 - 4 zeta) + 2 |Z| A (x, y terms as in the sq_scaled chain: 7u; the z term:
// subtraction, product, final fma: 3u), so  |q~ - q| <= u q (3.5 - 2 zeta + 1.56) + A sqrt(zeta); over zeta in [0, 1] and
// q <= qmax = nbins + 1, with a = A / (u qmax):  <= u qmax (5.06 + a^2 / 8)  (a <= 4; else 3.06 + a).  10 % margin on both.
// Returns the bound in bins (without the fixed-point grid term g_m).
inline double fast_guard_zf(int nbins, double hb, double gfrac)
{
    const double u = 1.0 / 16777216.0, qmax = (double)nbins + 1.0;
    const double a = hb * (gfrac + 1.0 / 16 + 1.0 / 128) / qmax;
    const double extra = a <= 4.0 ? a * a / 8.0 : a - 2.0;
    return 1.1 * u * qmax * (3.5 + extra + 1.56);
}

// TRI form of the tile kernel (general cells, rdf.hip tri_q): the pair vector in the orthogonalised lattice frame,
//   X = L00 (ix + c10 iy), Y = L11 iy, Z from f32 slab coordinates as in the ZF form (or L22 iz on the integer path),
// ix, iy the u32 differences of the FOLDED coordinates.  The x term: two conversions, the constant c10 and the fma leave
// |X~ - X| <= u (2 |X| + 3 a), a = |L10| / 2 in bins (the largest |L10 iy|); squared, scaled, summed: 9u X^2 + 6u a |X|;
// y term 7u Y^2, z term as in fast_guard_zf.  With zeta = Z^2 / T:
//   |q~ - q| <= u q (4.5 - 3 zeta + 1.56) + A sqrt(zeta) + 3 u a  <=  u qmax (6.06 + a_z^2 / 12 + 3 a / qmax)
// (a_z = A / (u qmax) <= 6; else 3.06 + a_z).  10 % margin.  Returns the bound in bins (without the grid term g_m).
inline double fast_guard_tri(int nbins, double hb, double gfrac, double l10_bins)
{
    const double u = 1.0 / 16777216.0, qmax = (double)nbins + 1.0;
    const double az = hb * (gfrac + 1.0 / 16 + 1.0 / 128) / qmax;
    const double extra = az <= 6.0 ? az * az / 12.0 : az - 3.0;
    return 1.1 * u * qmax * (4.5 + extra + 1.56 + 1.5 * fabs(l10_bins) / qmax);
}

}  // namespace amof
