// Matrix4x4::inverse (transform.rs:128-200) without dynamic indexing — round 6.
//
// The reference's Gauss-Jordan elimination chooses its pivot by a search over the rows and columns not used yet (the largest |element|, the LAST one
// in row-major scan order among equals), swaps two rows, eliminates, and at the end swaps columns back: every index is data.  Written that way for a
// GPU lane (16 elements in registers) each dynamic index is a chain of selects — 4 088 VALU instructions, 2 491 of them v_cndmask, per inverse — and
// AnimatedTransform::interpolate inverts the blended scale matrix at EVERY visit of a moving instance (primitive.rs:218-222, transform.rs:2106-2112).
//
// Same arithmetic, indices made static: every operation on an element (the division, the row scaling, `a[j][k] -= a[icol][k] * save`) depends on WHICH
// row is the pivot row and which column the pivot column, never on where they are stored.  So the matrix is kept in a PHYSICAL frame in which step s's
// pivot is moved to (s, s) — one conditional row exchange and one conditional column exchange per step — with two label arrays saying which logical
// row / column a physical one is.  The reference's own row swap (rows irow <-> icol) becomes an exchange of two row LABELS, its final column swaps an
// exchange of column labels, and the result is written out through the labels.  The pivot search compares (|element|, logical scan position) pairs,
// so ties fall as in the reference (uniform scales — diag(2, 2, 2, 1) — are all ties).  Element values, and therefore every rounding, are the
// reference's: tests/test_gpu_mat4_inverse.py holds the device function to the reference's compiled text (tests/golden/leaf_functions.npz) and to the
// oracle on tie-heavy families, bit for bit.  (A matrix whose remaining elements are ALL NaN at some step picks another pivot than the reference's
// (0, 0) default; both results are NaN throughout.)
#pragma once

#ifndef RSPT_M4_FN
#define RSPT_M4_FN RDEVN
#endif

namespace rspt {

RSPT_M4_FN void mat4_inverse(const float* src, float* out) {
    float a[4][4];
    int lr[4] = {0, 1, 2, 3}, lc[4] = {0, 1, 2, 3};   // logical row / column of a physical one
    int xr[4] = {0, 0, 0, 0}, xc[4] = {0, 0, 0, 0};   // indxr, indxc (logical)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) a[i][j] = src[4 * i + j];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        // choose pivot: the unused rows / columns are the physical ones s .. 3
        float big = 0.0f;
        int key = -1, pr = s, pc = s;
#pragma unroll
        for (int r = s; r < 4; r++)
#pragma unroll
            for (int c = s; c < 4; c++) {
                const float v = fabsf(a[r][c]);
                const int k = 4 * lr[r] + lc[c];
                const bool take = v > big || (v == big && k > key);   // `abs >= big` in scan order = the last of the largest
                big = take ? v : big; key = take ? k : key; pr = take ? r : pr; pc = take ? c : pc;
            }
        int irow = lr[s], icol = lc[s];
#pragma unroll
        for (int p = s + 1; p < 4; p++) { irow = pr == p ? lr[p] : irow; icol = pc == p ? lc[p] : icol; }
        xr[s] = irow; xc[s] = icol;
        // swap rows irow and icol: the pivot's row takes the label icol, the row that had it takes irow
#pragma unroll
        for (int r = s; r < 4; r++) lr[r] = r == pr ? icol : (lr[r] == icol ? irow : lr[r]);
        // the pivot to (s, s)
#pragma unroll
        for (int p = s + 1; p < 4; p++) {
            const bool sw = pr == p;
#pragma unroll
            for (int c = 0; c < 4; c++) { const float t = a[p][c]; a[p][c] = sw ? a[s][c] : t; a[s][c] = sw ? t : a[s][c]; }
            const int t = lr[p]; lr[p] = sw ? lr[s] : t; lr[s] = sw ? t : lr[s];
        }
#pragma unroll
        for (int p = s + 1; p < 4; p++) {
            const bool sw = pc == p;
#pragma unroll
            for (int r = 0; r < 4; r++) { const float t = a[r][p]; a[r][p] = sw ? a[r][s] : t; a[r][s] = sw ? t : a[r][s]; }
            const int t = lc[p]; lc[p] = sw ? lc[s] : t; lc[s] = sw ? t : lc[s];
        }
        // set m[icol][icol] to one by scaling its row, subtract the row from the others
        const float pivinv = 1.0f / a[s][s];
        a[s][s] = 1.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) a[s][c] *= pivinv;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (r == s) continue;
            const float save = a[r][s];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c == s) continue;
                a[r][c] -= a[s][c] * save;
            }
            // column s: `minv[j][icol] = 0.0; minv[j][icol] -= minv[icol][icol] * save` = 0 - p: -p for every p but +-0, where it is +0.  Written as that select because
            // hipcc 7.2 folds the literal subtraction into a NEGATION modifier of the instruction that reads it next ((0 - p) - q became (-p) - q, (0 - p) * y became
            // (-p) * y: -0 where the reference has +0 — 8 % of the small-integer matrices of tests/test_gpu_mat4_inverse.py differed in a zero's sign)
            const float p = a[s][s] * save;
            a[r][s] = p == 0.0f ? 0.0f : -p;
        }
    }
    // swap columns to reflect the permutation: exchanges of column labels, last step first
#pragma unroll
    for (int j = 3; j >= 0; j--)
#pragma unroll
        for (int c = 0; c < 4; c++) lc[c] = lc[c] == xr[j] ? xc[j] : (lc[c] == xc[j] ? xr[j] : lc[c]);
    // out[lr[r]][lc[c]] = a[r][c]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float row[4];
#pragma unroll
        for (int c = 0; c < 4; c++) row[c] = lr[0] == i ? a[0][c] : (lr[1] == i ? a[1][c] : (lr[2] == i ? a[2][c] : a[3][c]));
#pragma unroll
        for (int j = 0; j < 4; j++) out[4 * i + j] = lc[0] == j ? row[0] : (lc[1] == j ? row[1] : (lc[2] == j ? row[2] : row[3]));
    }
}

}  // namespace rspt
