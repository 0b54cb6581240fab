// Prints what amof_amd/csrc/guard_math.h computes for the inputs on stdin (tests/test_guard_math.py compares with numpy).
//   line "L r0 .. r8"            -> the nine entries of lower_factor(rows) and kappa_lower(L)
//   line "Z nbins hb gfrac"      -> fast_guard_zf(nbins, hb, gfrac)
#include <stdio.h>

#include "../../amof_amd/csrc/guard_math.h"

int main()
{
    char tag;
    while (scanf(" %c", &tag) == 1) {
        if (tag == 'L') {
            double r[9], L[9];
            for (int k = 0; k < 9; k++)
                if (scanf("%lf", &r[k]) != 1) return 1;
            amof::lower_factor(r, L);
            for (int k = 0; k < 9; k++) printf("%.17g ", L[k]);
            printf("%.17g\n", amof::kappa_lower(L));
        } else if (tag == 'Z') {
            int nbins;
            double hb, gfrac;
            if (scanf("%d %lf %lf", &nbins, &hb, &gfrac) != 3) return 1;
            printf("%.17g\n", amof::fast_guard_zf(nbins, hb, gfrac));
        } else {
            return 2;
        }
    }
    return 0;
}
