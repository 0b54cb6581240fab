// Test driver (CPU only): runs the text-ingest entry points of include/amof_hip.h over the files
// named on the command line, under AddressSanitizer + UBSan (built by tests/test_ingest_sanitized.py
// together with amof_amd/csrc/ingest.hip, which is plain host C++).  Prints one line per file.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/amof_hip.h"

int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        const char *path = argv[a];
        const bool is_cell = strstr(path, ".cell") != nullptr;
        if (is_cell) {
            int64_t rows = 0;
            int rc = amof_cp2k_cell_read(path, 0, nullptr, &rows);
            if (rc == 0 && rows > 0) {
                std::vector<double> cell((size_t)rows * 9);
                int64_t rows2 = 0;
                rc = amof_cp2k_cell_read(path, rows, cell.data(), &rows2);
                double s = 0;
                for (double v : cell) s += v;
                printf("%s cell rc=%d rows=%lld sum=%.17g\n", path, rc, (long long)rows2, s);
            } else {
                printf("%s cell rc=%d rows=%lld err=%s\n", path, rc, (long long)rows, rc ? amof_ingest_last_error() : "");
            }
            continue;
        }
        int64_t F = 0, N = 0;
        int rc = amof_xyz_scan(path, &F, &N);
        if (rc != 0) {
            printf("%s scan rc=%d err=%s\n", path, rc, amof_ingest_last_error());
            continue;
        }
        std::vector<double> pos((size_t)F * N * 3 + 1), lat((size_t)F * 9 + 1);
        std::vector<char> sym((size_t)N * 4 + 1);
        int32_t has = 0;
        for (int threads = 1; threads <= 3; threads += 2) {
            rc = amof_xyz_read(path, 0, F, 1, N, pos.data(), sym.data(), lat.data(), &has, threads);
            double s = 0;
            if (rc == 0)
                for (size_t k = 0; k < (size_t)F * N * 3; k++) s += pos[k];
            printf("%s read(threads=%d) rc=%d F=%lld N=%lld lattice=%d sum=%.17g %s\n", path, threads, rc, (long long)F,
                   (long long)N, (int)has, s, rc ? amof_ingest_last_error() : "");
        }
        if (F > 1) {   // strided subset
            rc = amof_xyz_read(path, F - 1, (F + 1) / 2, -2, N, pos.data(), sym.data(), nullptr, &has, 2);
            printf("%s read(reverse stride) rc=%d %s\n", path, rc, rc ? amof_ingest_last_error() : "");
        }
    }
    return 0;
}
