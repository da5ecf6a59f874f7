// cand_stats.h -- what the candidates' order statistics share (V4, mean_var_shift_polyA_check, reference
// adapted/detect/mvs.py:45-158, for ALL poly(A) candidates of a read: adapted/detect/combined.py:464): the result record and
// np.percentile's arithmetic.  The kernel is k_cand_stats2 (cand_stats2.h).  (The round-2 kernel that lived here -- four to five
// sweeps per array, an LDS atomic per element and level, 4 x slower -- is kept as tools/experiments/r05_pruned_variants.patch.)
#pragma once
#include "common.h"

#define CS_GROUP 10 // candidates per round (6 queries each on the slice itself)

struct CandStat { float fvar, fmean, fmed; int32_t ready; double q85, q15; };

// np.percentile (linear) from x_(hi) = vk and x_(hi - 1) = vkm1: the arithmetic of wave_percentile
static __device__ __forceinline__ void cs_pct_ranks(int n, double q100, int &lo, int &hi, double &g)
{
    const double vi = (double)(n - 1) * (q100 / 100.0);
    lo = (int)floor(vi);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    hi = lo + 1 < n - 1 ? lo + 1 : n - 1;
    g = vi - (double)lo;
}
static __device__ __forceinline__ double cs_pct_value(float vk, float vkm1, int lo, int hi, double g)
{
    const float a = (hi == lo) ? vk : vkm1, b = vk;
    const float diff = b - a;
    double r = (double)a + (double)diff * g;
    if (g >= 0.5) r = (double)b - (double)diff * (1.0 - g);
    return r;
}
