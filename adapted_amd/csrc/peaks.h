// peaks.h -- peak picking on the LLR traces, one wave per read.
//
//   T1 LLRTrace._trace_start_end                       reference adapted/detect/llr.py:135-142
//   P1 find_peaks_in_trace                             adapted/detect/llr.py:204-224
//   P2 correct_for_plateau                             adapted/detect/llr.py:145-177
//   P3 correct_for_split_peak                          adapted/detect/llr.py:180-201
//   A1 first candidate -> adapter_end                  adapted/detect/combined.py:180-188
//   P4 detect_full_polya_trace_peak_with_spike         adapted/detect/llr.py:406-479
//
// scipy.signal.find_peaks (third party, not in the reference tree) is restated from its
// published algorithm: strict local maxima with plateau midpoints -> optional minimum
// distance by descending height -> prominence (walk to the next higher sample on each side,
// lowest point on the way) -> width at rel_height by linear interpolation.  Here the 64 lanes
// of a wave evaluate 64 candidate maxima at once; the walks skip whole 64-point blocks using
// the per-block max/min the gains kernel left behind, and the minimum-distance rule is solved
// as its fixed point ("kept iff no kept higher neighbour within the distance") instead of a
// global sort.
#pragma once
#include "common.h"

#define DBLMAX 1.7976931348623157e308
extern __device__ int g_ablate;
extern __device__ unsigned long long g_dbg[ADP_NDBG]; // (debug tallies; 16-23: k_polya_peak in a -DADP_PHASE_TIMING build)

// P: `const double *` (the trace in global memory) or `const LDS double *` (a short trace staged in LDS: every walk below is a chain
// of dependent loads, a memory round trip each from HBM / L2)
template <class P>
struct TraceViewT {
    P x;                 // trace of one read (full-trace coordinates)
    P bmax;              // per-64 block max (NaN -> +inf) or nullptr
    P bmin;              // per-64 block min or nullptr
    int lo, hi;          // inclusive index bounds of the array find_peaks sees
    int sanitize;        // apply np.nan_to_num(nan=0) on load
};
typedef TraceViewT<const double *> TraceView;
typedef TraceViewT<const LDS double *> TraceViewL;

template <class TV>
static __device__ __forceinline__ double tv_get(const TV &t, int i)
{
    double v = t.x[i];
    if (t.sanitize) {
        if (v != v) v = 0.0;
        else if (__builtin_isinf(v)) v = v > 0 ? DBLMAX : -DBLMAX;
    }
    return v;
}

// Is full-trace index i the left edge of a strict local maximum (scipy _local_maxima_1d)?
// Returns the plateau midpoint or -1.
template <class TV>
static __device__ __forceinline__ int tv_peak_at(const TV &t, int i)
{
    if (i <= t.lo || i >= t.hi) return -1;
    double xi = tv_get(t, i);
    if (!(tv_get(t, i - 1) < xi)) return -1;
    int j = i + 1;
    while (j < t.hi && tv_get(t, j) == xi) j++;
    if (tv_get(t, j) < xi) return (i + j - 1) / 2;
    return -1;
}

// scipy's _peak_widths walks from the peak towards each side's prominence BASE -- the first sample, going away from the
// peak, that attains the side's minimum -- and stops there at the latest.  At rel_height = 1 the evaluation height
// x[p] - prominence is that minimum up to rounding, so whether the walk ends by value or at the base is decided by the last
// bit: the walks below carry the two side minima and stop on a sample EQUAL to its side's minimum as well as on one at or
// below the height (equivalent to carrying the base indices).

// ---- long walks, cooperatively ------------------------------------------------------------------------
// The dominant peak of a trace has nothing higher on either side: its prominence walk runs to the ends of the
// trace (hundreds of dependent steps in ONE lane while 63 wait), and its width walk is long too.  A lane whose
// private walk is not done within a budget hands the peak to the whole wave: 64 summary blocks (or the 64 samples
// of one block) are tested per step.
enum { WALK_PROM = 0, WALK_WIDTH = 1 };

template <int KIND>
static __device__ __forceinline__ bool walk_block_clean(double bmx, double bmn, double xp, double height, double smin)
{
    return KIND == WALK_PROM ? (bmx <= xp) : (bmn > height && bmn > smin && bmx <= xp);
}
template <int KIND>
static __device__ __forceinline__ bool walk_hit(double v, double xp, double height, double smin)
{
    return KIND == WALK_PROM ? !(v <= xp) : (!(height < v) || v == smin);
}

// nearest j in [limit, start], searching downwards, with walk_hit(x[j]); limit - 1 if none.  mn (WALK_PROM):
// minimum of the samples passed, i.e. those above j.  Uniform call (all lanes, same arguments).
template <int KIND, class TV>
static __device__ int coop_find_down(const TV &t, int start, int limit, double xp, double height, double &mn, double smin = 0.0)
{
    const int ln = lane_id();
    double lmn = xp;
    int i = start, found = limit - 1;
    while (i >= limit) {
        int b = i / SUMBLK, bstart = b * SUMBLK;
        if (t.bmax && i == bstart + SUMBLK - 1 && bstart >= limit) {
            // up to 64 whole blocks at once: lane l looks at block b - l
            const int bb = b - ln;
            const bool whole = bb >= 0 && bb * SUMBLK >= limit;
            const double bmx = whole ? t.bmax[bb] : 0.0, bmn = whole ? t.bmin[bb] : 0.0;
            const bool dirty = !whole || !walk_block_clean<KIND>(bmx, bmn, xp, height, smin);
            const unsigned long long m = __ballot(dirty);
            const int first = m ? __ffsll((long long)m) - 1 : 64;
            if (KIND == WALK_PROM && ln < first && bmn < lmn) lmn = bmn;
            b -= first; bstart = b * SUMBLK;
            i = bstart + SUMBLK - 1;
            if (first == 64 || i < limit) continue;
        }
        // the samples of block b within [limit, i]
        const int e = bstart + ln;
        const bool in = e >= limit && e <= i;
        const double v = in ? tv_get(t, e) : 0.0;
        const unsigned long long m = __ballot(in && walk_hit<KIND>(v, xp, height, smin));
        if (m) {
            found = bstart + 63 - __clzll((long long)m);
            if (KIND == WALK_PROM && in && e > found && v < lmn) lmn = v;
            break;
        }
        if (KIND == WALK_PROM && in && v < lmn) lmn = v;
        i = bstart - 1;
    }
    if (KIND == WALK_PROM) mn = wave_min(lmn);
    return found;
}

// nearest j in [start, limit], searching upwards; limit + 1 if none
template <int KIND, class TV>
static __device__ int coop_find_up(const TV &t, int start, int limit, double xp, double height, double &mn, double smin = 0.0)
{
    const int ln = lane_id();
    double lmn = xp;
    int i = start, found = limit + 1;
    while (i <= limit) {
        int b = i / SUMBLK, bstart = b * SUMBLK;
        if (t.bmax && i == bstart && bstart + SUMBLK - 1 <= limit) {
            const int bb = b + ln;
            const bool whole = bb * SUMBLK + SUMBLK - 1 <= limit;
            const double bmx = whole ? t.bmax[bb] : 0.0, bmn = whole ? t.bmin[bb] : 0.0;
            const bool dirty = !whole || !walk_block_clean<KIND>(bmx, bmn, xp, height, smin);
            const unsigned long long m = __ballot(dirty);
            const int first = m ? __ffsll((long long)m) - 1 : 64;
            if (KIND == WALK_PROM && ln < first && bmn < lmn) lmn = bmn;
            b += first; bstart = b * SUMBLK;
            i = bstart;
            if (first == 64 || i > limit) continue;
        }
        const int e = bstart + ln;
        const bool in = e >= i && e <= limit;
        const double v = in ? tv_get(t, e) : 0.0;
        const unsigned long long m = __ballot(in && walk_hit<KIND>(v, xp, height, smin));
        if (m) {
            found = bstart + __ffsll((long long)m) - 1;
            if (KIND == WALK_PROM && in && e < found && v < lmn) lmn = v;
            break;
        }
        if (KIND == WALK_PROM && in && v < lmn) lmn = v;
        i = bstart + SUMBLK;
    }
    if (KIND == WALK_PROM) mn = wave_min(lmn);
    return found;
}

// the walks of tv_prominence / tv_width with a step budget.  WALK_CHUNK samples are
// loaded per round trip to memory (the loads do not depend on the comparisons), then examined in walk order.
#ifndef WALK_CHUNK
#define WALK_CHUNK 4
#endif
// Result: PB_DONE (both walks ended: prom exact), PB_REJECT (a COMPLETED side's minimum already shows prominence < pmin: the prominence
// is xp - max(left_min, right_min) <= xp - either minimum, in floating point too, so whatever the other side holds the test
// pmin <= prominence fails -- the noise maxima of a rising or falling stretch, whose walk on the far side would run on to the end of
// the trace, leave here), or PB_LEFT / PB_RIGHT bits: that side's walk was cut short by the budget (its minimum so far is not final).
enum { PB_DONE = 0, PB_LEFT = 1, PB_RIGHT = 2, PB_REJECT = 4 };
template <class TV>
static __device__ int tv_prominence_budget(const TV &t, int p, int budget, double pmin, double &prom, double &lmin_out, double &rmin_out)
{
    const double xp = tv_get(t, p);
    double left_min = xp, right_min = xp;
    int more = 0;
    bool stopped = false;
    for (int i0 = p, steps = 0; !stopped && i0 >= t.lo; i0 -= WALK_CHUNK, steps += WALK_CHUNK) {
        if (steps >= budget) { more |= PB_LEFT; break; }
        double v[WALK_CHUNK];
#pragma unroll
        for (int u = 0; u < WALK_CHUNK; u++) v[u] = (i0 - u >= t.lo) ? tv_get(t, i0 - u) : 0.0;
#pragma unroll
        for (int u = 0; u < WALK_CHUNK; u++) {
            if (!stopped) {
                if (i0 - u < t.lo || !(v[u] <= xp)) stopped = true;
                else if (v[u] < left_min) left_min = v[u];
            }
        }
    }
    if (!(more & PB_LEFT) && !(pmin <= xp - left_min)) return PB_REJECT;
    stopped = false;
    for (int i0 = p, steps = 0; !stopped && i0 <= t.hi; i0 += WALK_CHUNK, steps += WALK_CHUNK) {
        if (steps >= budget) { more |= PB_RIGHT; break; }
        double v[WALK_CHUNK];
#pragma unroll
        for (int u = 0; u < WALK_CHUNK; u++) v[u] = (i0 + u <= t.hi) ? tv_get(t, i0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < WALK_CHUNK; u++) {
            if (!stopped) {
                if (i0 + u > t.hi || !(v[u] <= xp)) stopped = true;
                else if (v[u] < right_min) right_min = v[u];
            }
        }
    }
    if (!(more & PB_RIGHT) && !(pmin <= xp - right_min)) return PB_REJECT;
    prom = xp - (left_min > right_min ? left_min : right_min);
    lmin_out = left_min; rmin_out = right_min;
    return more;
}
template <class TV>
static __device__ bool tv_width_budget(const TV &t, int p, double prom, double rel, int budget, double &width, double lmin, double rmin)
{
    const double xp = tv_get(t, p);
    const double height = xp - prom * rel;
    // left: the first index i (descending from p, i > lo) with !(height < x[i]); lo if none
    int il = t.lo;
    {
        bool stopped = false;
        for (int i0 = p, steps = 0; !stopped && i0 > t.lo; i0 -= WALK_CHUNK, steps += WALK_CHUNK) {
            if (steps >= budget) return false;
            double v[WALK_CHUNK];
#pragma unroll
            for (int u = 0; u < WALK_CHUNK; u++) v[u] = (i0 - u > t.lo) ? tv_get(t, i0 - u) : 0.0;
#pragma unroll
            for (int u = 0; u < WALK_CHUNK; u++) {
                if (!stopped) {
                    if (i0 - u <= t.lo) { stopped = true; il = t.lo; }
                    else if (!(height < v[u]) || v[u] == lmin) { stopped = true; il = i0 - u; }
                }
            }
        }
    }
    int ir = t.hi;
    {
        bool stopped = false;
        for (int i0 = p, steps = 0; !stopped && i0 < t.hi; i0 += WALK_CHUNK, steps += WALK_CHUNK) {
            if (steps >= budget) return false;
            double v[WALK_CHUNK];
#pragma unroll
            for (int u = 0; u < WALK_CHUNK; u++) v[u] = (i0 + u < t.hi) ? tv_get(t, i0 + u) : 0.0;
#pragma unroll
            for (int u = 0; u < WALK_CHUNK; u++) {
                if (!stopped) {
                    if (i0 + u >= t.hi) { stopped = true; ir = t.hi; }
                    else if (!(height < v[u]) || v[u] == rmin) { stopped = true; ir = i0 + u; }
                }
            }
        }
    }
    double left_ip = (double)il, right_ip = (double)ir;
    { double xi = tv_get(t, il); if (xi < height) left_ip += (height - xi) / (tv_get(t, il + 1) - xi); }
    { double xi = tv_get(t, ir); if (xi < height) right_ip -= (height - xi) / (tv_get(t, ir - 1) - xi); }
    width = right_ip - left_ip;
    return true;
}

// Per lane: does the local maximum p (or -1: none) pass prominence >= pmin and width(rel) >= wmin?
// Uniform call.  Short walks run privately in each lane, long ones cooperatively, one peak at a time.
// (round 4, with the certain-reject exit of tv_prominence_budget: budget x chunk 48 x 8 -> 8 x 4; 96 000 reads, k_adapter_peak / k_polya_peak:
// 2.2 / 2.25 -> 1.1 / 1.57 ms at the preset's 16 k window, 2.33 / 5.65 -> 1.2 / 5.12 ms at the 200 k window; 8 x 8, 12 x 4, 16 x 8 within 5 %,
// 0 (every walk cooperative) 1.6 / 2.16, 96 x 8 3.4 / 2.9 -- tools/ab_libs.sh)
#ifndef WALK_BUDGET
#define WALK_BUDGET 8
#endif
template <class TV>
static __device__ bool wave_peak_ok(const TV &t, int p, double pmin, double wmin, double rel)
{
    const int ln = lane_id();
    double prom = 0.0, lmn = 0.0, rmn = 0.0; // prominence and the minima of the two sides (the values at the bases)
    bool have = p >= 0, done = true;
    int pb = PB_DONE;
    if (have) pb = tv_prominence_budget(t, p, WALK_BUDGET, pmin, prom, lmn, rmn);
    if (pb == PB_REJECT) have = false;
    unsigned long long todo = __ballot(have && pb != PB_DONE);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int pp = __shfl(p, src), sides = __shfl(pb, src);
        const double xp = tv_get(t, pp);
        double lmin = __shfl(lmn, src), rmin = __shfl(rmn, src); // (a side whose private walk ended keeps its minimum)
        bool rej = false;
        if (sides & PB_LEFT) { coop_find_down<WALK_PROM>(t, pp, t.lo, xp, 0.0, lmin); rej = !(pmin <= xp - lmin); } // (as PB_REJECT; uniform)
        if (!rej && (sides & PB_RIGHT)) coop_find_up<WALK_PROM>(t, pp, t.hi, xp, 0.0, rmin);
        if (ln == src) { if (rej) have = false; else { prom = xp - (lmin > rmin ? lmin : rmin); lmn = lmin; rmn = rmin; } }
    }
    bool cand = have && (pmin <= prom);
    double width = 0.0;
    done = true;
    if (cand) done = tv_width_budget(t, p, prom, rel, WALK_BUDGET, width, lmn, rmn);
    todo = __ballot(cand && !done);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int pp = __shfl(p, src);
        const double pr = __shfl(prom, src);
        const double sl = __shfl(lmn, src), sr = __shfl(rmn, src);
        const double xp = tv_get(t, pp);
        const double height = xp - pr * rel;
        double dummy;
        int il = coop_find_down<WALK_WIDTH>(t, pp, t.lo + 1, xp, height, dummy, sl);
        if (il < t.lo + 1) il = t.lo;
        int ir = coop_find_up<WALK_WIDTH>(t, pp, t.hi - 1, xp, height, dummy, sr);
        if (ir > t.hi - 1) ir = t.hi;
        double left_ip = (double)il, right_ip = (double)ir;
        { double xi = tv_get(t, il); if (xi < height) left_ip += (height - xi) / (tv_get(t, il + 1) - xi); }
        { double xi = tv_get(t, ir); if (xi < height) right_ip -= (height - xi) / (tv_get(t, ir - 1) - xi); }
        if (ln == src) width = right_ip - left_ip;
    }
    return cand && (wmin <= width);
}

// First peak (lowest index) of find_peaks(x[lo..hi], prominence=pmin, width=wmin, rel_height=rel),
// in full-trace coordinates, or -1.  Wave-cooperative: call from uniform control flow.
template <class TV>
static __device__ int wave_first_peak(const TV &t, double pmin, double wmin, double rel)
{
    const int ln = lane_id();
    for (int base = t.lo + 1; base < t.hi; base += 64) {
        int i = base + ln;
        int p = (i < t.hi) ? tv_peak_at(t, i) : -1;
        const bool ok = wave_peak_ok(t, p, pmin, wmin, rel);
        unsigned long long mk = __ballot(ok);
        if (mk) return __shfl(p, __ffsll((long long)mk) - 1);
    }
    return -1;
}

// ---------------------------------------------------------------- adapter end (P1 + P2 + P3 + A1)
// One read by one wave: the candidate in pooled units, or -1.  g / bmx / bmn: the read's trace and block summaries, in global memory or
// staged in LDS (P); se = T1's clip; s1 / s2 / nnan: the moments k_gains<1> left.
template <class P>
static __device__ int adapter_peak_read(P g, P bmx, P bmn, int n, int2 se, double s1, double s2, int nnan,
                                        double prominence, double rel_height, int width)
{
    const int ln = lane_id();
    const int cn = se.y - se.x; // clip = g[start:end]
    if (n < 3 || cn < 3) return -1;
    // np.nanstd(clip), clip = g[start:end]: k_gains<1> left the sum, sum of squares and NaN count of the whole
    // trace; the few points outside the clip (all <= 0, never NaN, except possibly g[end]) are taken out here.
    // (One-pass float64 moments instead of numpy's two passes: it only scales a threshold.)
    double o1 = 0.0, o2 = 0.0; int onan = 0;
    // (eight loads in flight per lane: these two sweeps cover most of the trace of a typical read)
    for (int i0 = 0; i0 < se.x; i0 += 512) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + ln; v[u] = i < se.x ? g[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { o1 += v[u]; o2 += v[u] * v[u]; }
    }
    for (int i0 = se.y; i0 < n; i0 += 512) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + ln; v[u] = i < n ? g[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { if (v[u] == v[u]) { o1 += v[u]; o2 += v[u] * v[u]; } else onan++; }
    }
    o1 = wave_sum(o1); o2 = wave_sum(o2); onan = wave_sum(onan);
    const int cnt = cn - (nnan - onan);
    const double sum = s1 - o1, sq = s2 - o2;
    const double mean = sum / (double)cnt;
    double var = (sq - sum * mean) / (double)cnt;
    if (var < 0.0) var = 0.0;
    double sd = sqrt(var);
    TraceViewT<P> tv{g, bmx, bmn, se.x, se.y - 1, 0};
    int peak = wave_first_peak(tv, prominence * sd, (double)width, rel_height);
    if (peak < 0) return -1;
    // P2 correct_for_plateau(trace, peak, s=10, t=0.9, window=500)
    {
        const int wn = min(peak + 500, n) - peak;
        const int nch = wn - 1;
        const double w0 = g[peak];
        int best = -1;
        for (int i = ln; i <= nch - 10; i += 64) {
            bool ok = g[peak + i + 9] > 0.9 * w0;
            for (int j = i; ok && j < i + 9; j++) ok = (g[peak + j + 1] - g[peak + j] >= 0.0);
            if (ok) best = i;
        }
        best = wave_max(best);
        if (best >= 0) peak += best + 9;
    }
    // P3 correct_for_split_peak: find_peaks(window, width=10, prominence=1.0)[0]
    {
        const int wn = min(peak + 500, n) - peak;
        TraceViewT<P> tw{g, nullptr, nullptr, peak, peak + wn - 1, 0};
        int pk = wave_first_peak(tw, 1.0, 10.0, 0.5);
        if (pk >= 0 && g[pk] >= 0.9 * g[peak]) peak = pk;
    }
    return peak;
}

// grid = n_reads waves.  adapter_idx[r] = candidate in pooled units, or -1.
// (Round 4, measured and dropped: traces of at most 2048 points walked from an LDS copy by a launch of their own -- 2.57 against
// 2.68 ms at the 16 k window, 3.2 against 2.65 on heavy-tailed lengths: the walks are bound by their serial steps, not by where
// the samples come from.  tools/experiments/r05_pruned_variants.patch)
__global__ void __launch_bounds__(64) k_adapter_peak(const double *__restrict__ trace, const int32_t *__restrict__ nvalid, int Lp,
                                                     const double *__restrict__ bmax, const double *__restrict__ bmin, int nsum,
                                                     const int2 *__restrict__ t1, int mbsize, const MbState *__restrict__ mbs,
                                                     double prominence, double rel_height, int width,
                                                     int32_t *__restrict__ adapter_idx, const double *__restrict__ gstat)
{
    const int r = blockIdx.x;
    const int ln = lane_id();
    if (mbs[r / mbsize].status != ADP_MB_OK) { if (ln == 0) adapter_idx[r] = -1; return; }
    const int n = nvalid[r];
    int result = -1;
    if (n >= 3) {
        const double *g = trace + (size_t)r * Lp;
        const double *bx = bmax + (size_t)r * nsum, *bn = bmin + (size_t)r * nsum;
        const int2 se = t1[r];
        const double s1 = gstat[3 * r], s2 = gstat[3 * r + 1];
        const int nnan = (int)gstat[3 * r + 2];
        result = adapter_peak_read<const double *>(g, bx, bn, n, se, s1, s2, nnan, prominence, rel_height, width);
    }
    if (ln == 0) adapter_idx[r] = result;
}

// ---------------------------------------------------------------- poly(A) end (P4)
// 2-bit state per local maximum, by ORDINAL in the read's index-ordered list of maxima; transitions only clear
// bits (one LDS atomic and): undecided 3 -> kept 2 / removed 1
#ifndef PK_STEP
#define PK_STEP 16 // kept maxima examined per step of step 4
#endif
#ifndef PK_PREFIX
#define PK_PREFIX 224 // maxima of the first attempt (one round of step 2; 96 000 reads at the 200 k window: 5.2 ms without the prefix, 2.88 with 896, 2.50 with 448, 2.30 with 224, 2.38 with 112); ADP_ABLATE bit 2^24: the whole list at once, as before
#endif
#define PST_NONE 0u
#define PST_REMOVED 1u
#define PST_KEPT 2u
#define PST_UNDECIDED 3u

// persistent grid of waves.  pk[Lp/2+1] int32 per READ (list of maxima, pre-filled by k_gains; ends up holding
// the kept ones), mk[Lp/2+1] uint32 per SLOT (work list of undecided maxima: ordinal << 8 | neighbour mask);
// the states live in dynamic LDS: ((Lp/2+1) + 8) / 16 + 2 words per wave.
//
// Strict local maxima are at least two samples apart, so at most four of them lie within the minimum distance
// (|dp| <= 9) on either side of a maximum: the neighbourhood is the ordinals k-4 .. k+4, whatever the positions.
__global__ void __launch_bounds__(64, 8) k_polya_peak(const double *__restrict__ trace, const int32_t *__restrict__ nvalid, int Lp,
                                                   const double *__restrict__ bmax, const double *__restrict__ bmin, int nsum,
                                                   const int32_t *__restrict__ adapter_idx, int n_reads, int mbsize,
                                                   const MbState *__restrict__ mbs, int32_t *__restrict__ pk_all,
                                                   uint32_t *__restrict__ mk_all, int32_t *__restrict__ polya_idx,
                                                   const int32_t *__restrict__ npk_all, const double *__restrict__ pkv_all = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t stw_raw[];
    LDS uint32_t *stw = (LDS uint32_t *)stw_raw; // ordinal k is slot k + 4 (slots 0..3 stay PST_NONE)
    const int ln = lane_id();
    const int half = Lp / 2 + 1;
    uint32_t *mk = mk_all + (size_t)blockIdx.x * half;
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        int32_t *pk = pk_all + (size_t)r * half; // per-read list, pre-filled by k_gains
        int result = 0;
        const int n = nvalid[r];
        const bool active = mbs[r / mbsize].status == ADP_MB_OK && adapter_idx[r] >= 0 && n >= 3;
        if (active) {
            const double *g = trace + (size_t)r * Lp;
            TraceView tv{g, bmax + (size_t)r * nsum, bmin + (size_t)r * nsum, 0, n - 1, 1};
            // 1. all local maxima, in index order: normally found by k_gains while the trace tile was in LDS;
            //    recounted here (one load per lane, neighbours by shuffle) when a plateau was met
            int npk = npk_all[r];
            const bool recount = npk < 0;
            // heights of the listed maxima as k_gains wrote them (none after a recount: gathered from the trace then)
            const double *pkv = (pkv_all && !recount) ? pkv_all + (size_t)r * half : nullptr;
            if (recount) npk = 0;
            double carry = 0.0; // value at base - 1
            __syncthreads();
            for (int base0 = 0; recount && base0 < n; base0 += 256) {
                double vv[4], ee[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { // four tiles in flight
                    const int i = base0 + u * 64 + ln;
                    vv[u] = (i < n) ? tv_get(tv, i) : 0.0;
                    ee[u] = (ln == 63 && i + 1 < n) ? tv_get(tv, i + 1) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int base = base0 + u * 64;
                    if (base >= n) break;
                    const int i = base + ln;
                    const double v = vv[u];
                    double prev = __shfl_up(v, 1);
                    if (ln == 0) prev = carry;
                    double next = __shfl_down(v, 1);
                    if (ln == 63) next = ee[u];
                    carry = __shfl(v, 63);
                    int p = -1;
                    if (i > 0 && i < n - 1 && prev < v) {
                        if (next < v) p = i;
                        else if (next == v) { // plateau: walk to its end (rare on float64 traces)
                            int j = i + 1;
                            while (j < n - 1 && tv_get(tv, j) == v) j++;
                            if (tv_get(tv, j) < v) p = (i + j - 1) / 2;
                        }
                    }
                    unsigned long long m = __ballot(p >= 0);
                    if (p >= 0) pk[npk + __popcll(m & ((1ull << ln) - 1ull))] = p;
                    npk += __popcll(m);
                }
            }
            if (g_ablate & 256) npk = 0;
            // Only the first two survivors in index order are wanted, and they lie near the front of almost every trace: steps 2-4 run
            // on a PREFIX of the list first (the first wn maxima; the maxima behind it read as undecided, so a state that depends on
            // them stays undecided -- whatever is decided is final: the fixed point is unique).  The answer stands when the second
            // survivor was found among maxima in front of the first undecided one; otherwise the steps run again on the whole list.
            int p0 = -1, p1 = -1;
            for (int attempt = 0; attempt < 2; attempt++) {
            const int wn = (attempt == 0 && npk > PK_PREFIX && !(g_ablate & 16777216)) ? PK_PREFIX : npk;
            __syncthreads();
            for (int w = ln; w < (wn + 8 + 15) / 16 + 1; w += 64) stw[w] = 0;
            __syncthreads();
            if (ln < 4 && wn + ln < npk) { // (ordinals wn .. wn + 3: neighbours of the prefix's last maxima)
                const uint32_t sl = (uint32_t)(wn + ln + 4);
                __hip_atomic_fetch_or(&stw[sl >> 4], PST_UNDECIDED << ((sl & 15u) * 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // 2. per maximum: which of the up to 8 maxima within +-9 samples have a higher priority
            //    (scipy _select_by_peak_distance, distance = 10 -> |dp| < 10; equal heights: later index first).
            //    56 maxima per step (lanes 4..59), their neighbours by shuffle.  A maximum without any is kept
            //    at once; the others form the work list of step 3.
            int nund = 0;
            // (four steps' positions, then their four trace values, are requested together: a step is two dependent round
            // trips to memory -- the list entry, then the trace value it points at -- and a read has up to 180 steps)
            for (int base4 = 0; base4 < wn; base4 += 56 * 4) {
                int pp[4];
                double vv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = base4 + 56 * u - 4 + ln;
                    const bool valid = k >= 0 && k < npk;
                    pp[u] = valid ? pk[k] : (k < 0 ? -0x40000000 : 0x40000000);
                }
                if (pkv) { // (uniform) positions and heights are independent loads: one round trip per step instead of two
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = base4 + 56 * u - 4 + ln;
                        const bool valid = k >= 0 && k < npk;
                        vv[u] = valid ? pkv[k] : 0.0;
                    }
                } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = base4 + 56 * u - 4 + ln;
                    const bool valid = k >= 0 && k < npk;
                    vv[u] = valid ? tv_get(tv, pp[u]) : 0.0;
                }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int base = base4 + 56 * u;
                    if (base >= wn) break;
                    const int k = base - 4 + ln;
                    const bool valid = k >= 0 && k < npk;
                    const int p = pp[u];
                    const double v = vv[u];
                    // forward neighbours k+j by shuffle; the backward relation of k to k-j is the forward one seen from
                    // k-j with the comparison reversed (the sanitised values hold no NaN), so it travels as one bit
                    uint32_t mask = 0, lower = 0; // lower bit j-1: k+j lies within 9 samples and is LOWER than k
#pragma unroll
                    for (int j = 1; j <= 4; j++) {
                        const int pf = __shfl_down(p, j); const double vf = __shfl_down(v, j);
                        const bool within = pf - p <= 9;
                        if (within && vf >= v) mask |= 1u << (j - 1);
                        if (within && !(vf >= v)) lower |= 1u << (j - 1);
                    }
#pragma unroll
                    for (int j = 1; j <= 4; j++) {
                        const uint32_t lb = (uint32_t)__shfl_up((int)lower, j);
                        if (lb >> (j - 1) & 1u) mask |= 1u << (4 + j - 1); // k-j is within 9 and k is lower than it
                    }
                    const bool out = valid && ln >= 4 && ln < 60 && k < wn;
                    if (out) {
                        const uint32_t sl = (uint32_t)(k + 4);
                        __hip_atomic_fetch_or(&stw[sl >> 4], (mask ? PST_UNDECIDED : PST_KEPT) << ((sl & 15u) * 2u), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const bool und = out && mask;
                    const unsigned long long m = __ballot(und);
                    if (und) mk[nund + __popcll(m & ((1ull << ln) - 1ull))] = ((uint32_t)k << 8) | mask;
                    nund += __popcll(m);
                }
            }
            __syncthreads();
            // 3. fixed point of "kept iff no kept higher-priority neighbour"
            if (g_ablate & 512) nund = 0;
            while (nund > 0) {
                int w = 0;
                for (int base4 = 0; base4 < nund; base4 += 256) { // (four list loads in flight; the rewritten list stays
                    uint32_t ee[4];                                 //  below what has been read)
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int idx = base4 + 64 * u + ln; ee[u] = (idx < nund) ? mk[idx] : 0u; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int base = base4 + 64 * u;
                        if (base >= nund) break;
                        const int idx = base + ln;
                        const uint32_t e = ee[u];
                        bool pending = false;
                        if (idx < nund) {
                            const uint32_t k = e >> 8;
                            // states of the ordinals k-4 .. k+4 = slots k .. k+8
                            const unsigned long long W =
                                (((unsigned long long)stw[(k >> 4) + 1] << 32) | stw[k >> 4]) >> ((k & 15u) * 2u);
                            bool kept_nb = false;
                            uint32_t b = e & 255u;
                            while (b) {
                                const int o = __ffs(b) - 1; b &= b - 1;
                                const int f = o < 4 ? 5 + o : 7 - o; // k+1..k+4 -> fields 5..8 ; k-1..k-4 -> fields 3..0
                                const uint32_t sn = (uint32_t)(W >> (2 * f)) & 3u;
                                if (sn == PST_KEPT) kept_nb = true;
                                else if (sn == PST_UNDECIDED) pending = true;
                            }
                            const uint32_t sl = k + 4u;
                            if (kept_nb) {
                                __hip_atomic_fetch_and(&stw[sl >> 4], ~(2u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                pending = false;
                            } else if (!pending) {
                                __hip_atomic_fetch_and(&stw[sl >> 4], ~(1u << ((sl & 15u) * 2u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        const unsigned long long m = __ballot(pending);
                        if (pending) mk[w + __popcll(m & ((1ull << ln) - 1ull))] = e;
                        w += __popcll(m);
                        __syncthreads();
                    }
                }
                if (w == nund) break; // (a prefix: what is left waits for maxima behind it)
                nund = w;
            }
            __syncthreads();
            // kept maxima in front of the first undecided one, in index order, into the work list's place (it is dead by now)
            int32_t *kl = reinterpret_cast<int32_t *>(mk);
            int nkept = 0, first_und = wn;
            for (int base4 = 0; base4 < wn && first_und == wn; base4 += 256) { // (four loads in flight)
                int pq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int k = base4 + 64 * u + ln; pq[u] = (k < wn) ? pk[k] : -1; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = base4 + 64 * u + ln;
                    uint32_t stt = PST_NONE;
                    if (k < wn) {
                        const uint32_t sl = (uint32_t)(k + 4);
                        stt = (stw[sl >> 4] >> ((sl & 15u) * 2u)) & 3u;
                    }
                    const unsigned long long mu = __ballot(stt == PST_UNDECIDED);
                    if (mu && first_und == wn) first_und = base4 + 64 * u + __ffsll((long long)mu) - 1;
                    const bool kp = stt == PST_KEPT && k < first_und;
                    const unsigned long long m = __ballot(kp);
                    if (kp) kl[nkept + __popcll(m & ((1ull << ln) - 1ull))] = pq[u];
                    nkept += __popcll(m);
                }
            }
            __syncthreads();
            // 4. survivors in index order: prominence >= 1, width(rel 0.5) >= 10; first two
            // survivors, compacted in index order (LDS state scan), then 64 candidates per step
            p0 = -1; p1 = -1;
            // (16 candidates per step: the second survivor is among the first 32 kept maxima of almost every read, and
            // every candidate with a long walk costs the whole wave a cooperative scan)
            for (int base = 0; base < nkept && p1 < 0 && !(g_ablate & 1024); base += PK_STEP) {
                int k = base + ln;
                int i = (ln < PK_STEP && k < nkept) ? kl[k] : -1;
                const bool ok = wave_peak_ok(tv, i, 1.0, 10.0, 0.5);
                unsigned long long m = __ballot(ok);
                while (m && p1 < 0) {
                    int f = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    int pi = __shfl(i, f);
                    if (p0 < 0) p0 = pi; else p1 = pi;
                }
            }
            if (p1 >= 0 || first_und >= npk) break; // (else: the prefix did not settle it)
            }
            // 5. spike heuristics on the UN-sanitised trace
            if (p0 >= 0 && p1 < 0) result = p0;
            else if (p0 >= 0) {
                double h0 = g[p0], h1 = g[p1];
                if (h1 > h0) result = p1;
                else if (h1 < h0 * 0.5) result = p0;
                else {
                    // idx_min = argmin(g[p0:p1]) (first minimum; a NaN wins)
                    double mv = __builtin_inf(); int mi = 0x7fffffff; int nan_i = 0x7fffffff;
                    for (int i = p0 + ln; i < p1; i += 64) {
                        double v = g[i];
                        if (v != v) nan_i = min(nan_i, i);
                        else if (v < mv) { mv = v; mi = i; }
                    }
                    nan_i = wave_min(nan_i);
                    double gm = wave_min(mv);
                    int cand = (mv == gm) ? mi : 0x7fffffff;
                    cand = wave_min(cand);
                    int idx_min = (nan_i != 0x7fffffff) ? nan_i : cand;
                    if (idx_min == 0x7fffffff) idx_min = p0; // all +inf: np.argmin -> 0
                    int cnt = p1 - idx_min;
                    // scipy.stats.linregress r-value
                    double sx = 0, sy = 0;
                    for (int i = idx_min + ln; i < p1; i += 64) { sx += (double)i; sy += g[i]; }
                    sx = wave_sum(sx); sy = wave_sum(sy);
                    double xm = sx / cnt, ym = sy / cnt;
                    double sxx = 0, sxy = 0, syy = 0;
                    for (int i = idx_min + ln; i < p1; i += 64) {
                        double dx = (double)i - xm, dy = g[i] - ym;
                        sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
                    }
                    sxx = wave_sum(sxx); sxy = wave_sum(sxy); syy = wave_sum(syy);
                    double inv = 1.0 / (double)cnt;
                    double ssxm = sxx * inv, ssxym = sxy * inv, ssym = syy * inv;
                    double rr;
                    if (ssxm == 0.0 || ssym == 0.0) rr = 0.0;
                    else { rr = ssxym / sqrt(ssxm * ssym); if (rr > 1.0) rr = 1.0; else if (rr < -1.0) rr = -1.0; }
                    result = (rr * rr >= 0.99) ? p1 : 0;
                }
            }
            __syncthreads();
        }
        if (ln == 0) polya_idx[r] = result;
    }
}
