/* oracle/knn_brute.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Brute-force K=1 nearest neighbour, D=3, squared L2 -- restates the published algorithm of
 * chamferdist.knn_points (PyTorch3D-derived knn_cpu.cpp): for every query scan all reference
 * points in index order, keep the first strict minimum.  chamferdist is an un-vendored,
 * un-pinned dependency of the reference (README.md:17-19; loss/losses.py:3,57): PARITY UNPINNED.
 * Distance is evaluated as ((dx*dx + dy*dy) + dz*dz) in fp32 without fused multiply-add
 * (compile with -ffp-contract=off) so the HIP kernel can reproduce it bit for bit.
 */
#include <stdint.h>
#include <float.h>

void knn1_brute(const float* p1, int64_t n1, const float* p2, int64_t n2,
                float* dists, int64_t* idx)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        const float x = p1[3 * i], y = p1[3 * i + 1], z = p1[3 * i + 2];
        float best = FLT_MAX;
        int64_t bi = -1;
        for (int64_t j = 0; j < n2; ++j) {
            const float dx = x - p2[3 * j], dy = y - p2[3 * j + 1], dz = z - p2[3 * j + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bi = j; }
        }
        dists[i] = best;
        idx[i] = bi;
    }
}
