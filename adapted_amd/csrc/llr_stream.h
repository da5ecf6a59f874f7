// llr_stream.h -- D1 (clip + normalise + mean-pool), sequential float64 cumulative sums with
// checkpoints, and the LLR gains trace (G1/G2).
//
//   D1  downscale_signal(normalize_signal(batch[:, :T])[:, min_obs:], ds)
//       reference adapted/detect/downscale.py:4-41, adapted/detect/normalize.py:25-63,
//       adapted/detect/combined.py:128-140.  float32, numpy's pairwise order inside a pool block.
//   G1  c_llr_trace -> _gains(0, n-1, c, c2, 5, 5)      adapted/detect/_c_llr.pyx:202-236, 67-88
//   G2  c_llr_trace_gains -> _gains(a, n-1, c, c2, 1, 1) adapted/detect/_c_llr.pyx:176-199
//       var_c                                            adapted/detect/_c_llr.pyx:23-37
//       (the reference's libc log is replaced by log_cr.h's log_1ulp_fast: < 1 ULP like glibc's, 20 float64 operations
//       against ~75 of the device library's; the correctly rounded log_cr_fast, 45, stays for the whole-window term and the trace API)
//
// np.cumsum is a strictly sequential float64 recurrence and the pass-2 trace next to the
// adapter boundary is pure cumulative-sum rounding noise, so the recurrence is kept bit-exact:
// one LANE per read walks the pooled signal once (64 reads per wave) and stores the running
// sums every CK points; the gains kernel (one WAVE per read) restarts from those checkpoints,
// so every lane reproduces the reference's partial sums exactly while 64 split points are
// evaluated in parallel.
#pragma once
#include <type_traits>
#include "common.h"
#include "log_cr.h"
#include "wave_stats.h"

__device__ const double g_logcr_table[3 * LOGCR_N] = LOGCR_TABLE;

// ---------------------------------------------------------------- D1
#define NP_TILE 512
// grid = n_reads blocks of 256 threads; dynamic LDS = NP_TILE * ds floats
// ranges != nullptr (CNN fallback C4): per-read pooled region [ranges[2r], min(ranges[2r+1], T, full_len))
// tails_nan (ADP_TAILS_NAN): the row is NaN padding from full_len[r] on -- pooled blocks that reach into it are NaN without
// being read
// SP (the start-peak scan of K1 rides this pass, reference adapted/detect/start_peak.py:26-52): the tile then holds the RAW
// samples (normalised on their way into the pooled sums: the same operations on the same values) and every complete pooled
// block also yields the mean of its raw samples -- start_peak.py's `pooled` when both pooling factors agree and min_obs_adapter
// is a multiple of them -- which is compared with the read's start-peak maximum (SpHead, left by k_sp_head): the first block
// above it inside [a2, e0), and the first raw sample above the open-pore level before op_end.  k_sp_tail finishes the row.
struct SpHead {
    int32_t valid, max_idx, a2, e0;   // K1 state after the head: still valid, arg-max of pooled[off1:spmax], scan range [a2, e0)
    float mx;                         // max(pooled[off1:spmax]) (NaN if a NaN was there)
    int32_t op_end;                   // raw samples [0, op_end) are searched for the open pore (min(full_len, m) // ds: the quirk)
    int32_t op_head, op_body, hit;    // first raw index above the level in the head / in this pass's range; first block above mx here
    int32_t pad;
};

// NQ > 0 (= 16-byte loads per thread and tile, NP_TILE * ds / 1024): the loads of tile t + 1 are issued before tile t is pooled
// and stay in flight across its two barriers (NQ = 0: load, barrier, pool, barrier)
template <class SIG, bool SP = false, int NQ = 0>
__global__ void __launch_bounds__(256) k_norm_pool(SIG sig, int m, int T, int off, int ds, int L, int Lp,
                                                   int mbsize, const MbState *__restrict__ mbs,
                                                   float *__restrict__ down, int32_t *__restrict__ nvalid,
                                                   const int64_t *__restrict__ ranges, const int32_t *__restrict__ full_len,
                                                   int tails_nan = 0, SpHead *__restrict__ sp_head = nullptr, float sp_thr = 0.f)
{
    extern __shared__ __attribute__((aligned(16))) float tile[]; // (16-byte aligned: the static words in front of it would leave every float4 access of the tile misaligned -- slow, not wrong)
    __shared__ int s_nan;
    __shared__ __attribute__((aligned(16))) int s_sp[2];
    const int r = blockIdx.x;
    const MbState st = mbs[r / mbsize];
    if (st.status != ADP_MB_OK) { if (threadIdx.x == 0) nvalid[r] = 0; return; }
    const float med = st.med, mad = st.mad, lo = st.lo, hi = st.hi;
    int Lseg = T - off; // > 0 guaranteed by the host
    if (ranges) {
        long long a = ranges[2 * r], b = ranges[2 * r + 1];
        long long lim = full_len[r] < T ? full_len[r] : T;
        if (b > lim) b = lim;
        off = (int)a;
        Lseg = (int)(b > a ? b - a : 0);
        L = (Lseg + ds - 1) / ds;
        if (L > Lp) L = Lp;
    }
    const typename SIG::Row row = sig.row(r, m) + off;
    if (threadIdx.x == 0) s_nan = 0;
    int my_nan = 0;
    // start-peak state of this read (SP)
    float sp_mx = 0.f;
    int sp_a2 = 0, sp_e0 = 0, sp_op_end = 0, sp_hit = 0x7fffffff, sp_op = 0x7fffffff;
    const int sp_full = (T - off) / ds;      // complete pooled blocks of this pass: the ragged last one is left to k_sp_tail
    const int sp_pb0 = off / ds;             // pooled index (start_peak.py's) of this pass's block 0
    if (SP) {
        const SpHead hd = sp_head[r];
        sp_mx = hd.mx; sp_a2 = hd.valid ? hd.a2 : 0; sp_e0 = hd.valid ? hd.e0 : 0; sp_op_end = hd.op_end;
        if (threadIdx.x < 2) s_sp[threadIdx.x] = 0x7fffffff;
    }
    // pooled blocks worth computing: all of them, or only those that end before the padding starts
    int L_ok = L;
    if (tails_nan && !ranges) {
        long long lv = (long long)full_len[r] - off; // valid samples of the segment
        if (lv < Lseg) { if (lv < 0) lv = 0; L_ok = (int)(lv / ds); }
    }
    const int tile_n = NP_TILE * ds; // samples per tile (NP_TILE pooled outputs, 2 per thread)
    const bool vec = row.vec_ok() && ((tile_n & 3) == 0);
    // every sample of the minibatch is divided by the same MAD: one IEEE division for its reciprocal, then three
    // operations per sample that return the bits of the IEEE quotient (fdiv_shared, common.h)
    const bool fast_div = fdiv_ok(mad);
    const float rmad = 1.0f / mad;
    auto norm1 = [&](float c) {
        c = c < lo ? lo : c; // np.clip; NaN stays NaN
        c = c > hi ? hi : c;
        return fast_div ? fdiv_shared(c - med, mad, rmad) : (c - med) / mad;
    };
    const bool use_pf = NQ > 0 && vec && tile_n / 4 == NQ * 256;
    float4 pre[NQ > 0 ? NQ : 1];
    auto fetch = [&](int tb_) { // the raw samples of tile tb_ into registers (positions behind the segment: zeros, never used as samples)
        const int base_ = tb_ * ds;
        const typename SIG::Row rowb = row + base_;
#pragma unroll
        for (int u = 0; u < (NQ > 0 ? NQ : 1); u++) {
            const int q = threadIdx.x + u * 256;
            const int idx = base_ + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx + 3 < Lseg) v = rowb.f4s(q);
            else {
                if (idx < Lseg) v.x = row[idx];
                if (idx + 1 < Lseg) v.y = row[idx + 1];
                if (idx + 2 < Lseg) v.z = row[idx + 2];
            }
            pre[u] = v;
        }
    };
    if (use_pf && L_ok > 0 && L > 0) fetch(0);
    for (int tb = 0; tb < L; tb += NP_TILE) {
        const int base = tb * ds;
        if (tb >= L_ok) { // the padding: NaN blocks
            for (int jj = threadIdx.x; jj < NP_TILE; jj += 256) if (tb + jj < L) { down[(size_t)r * Lp + tb + jj] = __builtin_nanf(""); my_nan++; }
            continue;
        }
        __syncthreads();
        if (use_pf) {
#pragma unroll
            for (int u = 0; u < (NQ > 0 ? NQ : 1); u++) {
                const int q = threadIdx.x + u * 256;
                const int idx = base + 4 * q;
                float4 v = pre[u];
                if (!SP) { // normalised on the way into the tile; the pad behind the segment stays zero (np.pad)
                    v.x = idx < Lseg ? norm1(v.x) : 0.f; v.y = idx + 1 < Lseg ? norm1(v.y) : 0.f;
                    v.z = idx + 2 < Lseg ? norm1(v.z) : 0.f; v.w = idx + 3 < Lseg ? norm1(v.w) : 0.f;
                }
                reinterpret_cast<float4 *>(tile)[q] = v;
            }
        } else if (vec) {
            const typename SIG::Row rowb = row + base;
            for (int q = threadIdx.x; q < tile_n / 4; q += 256) {
                const int idx = base + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f); // np.pad(..., mode="constant") for a ragged tail
                if (SP) { // raw samples; a sample behind the segment is marked (its normalised value is the pad's zero)
                    const float padv = -__builtin_inff();
                    v = make_float4(padv, padv, padv, padv);
                    if (idx + 3 < Lseg) v = rowb.f4s(q);
                    else {
                        if (idx < Lseg) v.x = row[idx];
                        if (idx + 1 < Lseg) v.y = row[idx + 1];
                        if (idx + 2 < Lseg) v.z = row[idx + 2];
                    }
                } else if (idx + 3 < Lseg) { v = rowb.f4s(q); v.x = norm1(v.x); v.y = norm1(v.y); v.z = norm1(v.z); v.w = norm1(v.w); }
                else {
                    if (idx < Lseg) v.x = norm1(row[idx]);
                    if (idx + 1 < Lseg) v.y = norm1(row[idx + 1]);
                    if (idx + 2 < Lseg) v.z = norm1(row[idx + 2]);
                }
                reinterpret_cast<float4 *>(tile)[q] = v;
            }
        } else {
            for (int i = threadIdx.x; i < tile_n; i += 256) {
                int idx = base + i;
                if (SP) tile[i] = (idx < Lseg) ? row[idx] : -__builtin_inff();
                else tile[i] = (idx < Lseg) ? norm1(row[idx]) : 0.0f;
            }
        }
        __syncthreads();
        if (use_pf) { const int nx = tb + NP_TILE; if (nx < L && nx < L_ok) fetch(nx); }
        for (int jj = threadIdx.x; jj < NP_TILE; jj += 256) {
            const int j = tb + jj;
            if (j < L) {
                const float *p = tile + jj * ds;
                float s;
                if (SP) {
                    // (only the ragged last block holds pad marks: -inf never is a sample's value there, the range test says so)
                    const int nin = Lseg - j * ds; // samples of this block inside the segment (>= ds for a complete block)
                    if (nin >= ds) s = pw_leaf_f32(ds, [&](int k) { return norm1(p[k]); });
                    else s = pw_leaf_f32(ds, [&](int k) { return k < nin ? norm1(p[k]) : 0.0f; });
                    if (j < sp_full) {
                        const int g = sp_pb0 + j;
                        if (g >= sp_a2 && g < sp_e0) {
                            const float v = pw_leaf_f32(ds, [&](int k) { return p[k]; }) / (float)ds;
                            if (v > sp_mx && g < sp_hit) sp_hit = g;
                        }
                        const int i0 = off + j * ds; // raw index of the block's first sample
                        if (i0 < sp_op_end && sp_op == 0x7fffffff) {
                            for (int k = 0; k < ds; k++) if (i0 + k < sp_op_end && p[k] > sp_thr) { sp_op = i0 + k; break; }
                        }
                    }
                } else s = pw_leaf_f32(ds, [&](int k) { return p[k]; });
                float pooled = s / (float)ds;
                down[(size_t)r * Lp + j] = pooled;
                if (pooled != pooled) my_nan++;
            }
        }
    }
    if (SP) {
        sp_hit = wave_min(sp_hit); sp_op = wave_min(sp_op);
        if (lane_id() == 0) { if (sp_hit != 0x7fffffff) atomicMin(&s_sp[0], sp_hit); if (sp_op != 0x7fffffff) atomicMin(&s_sp[1], sp_op); }
    }
    my_nan = wave_sum(my_nan);
    if (lane_id() == 0 && my_nan) atomicAdd(&s_nan, my_nan);
    __syncthreads();
    if (threadIdx.x == 0) {
        nvalid[r] = L - s_nan;
        if (SP) { sp_head[r].hit = s_sp[0]; sp_head[r].op_body = s_sp[1]; }
    }
}

// ---------------------------------------------------------------- cumulative sums
// one lane per read.  ck[r][q] = (c, c2) BEFORE pooled sample q*CK; tail[r] = (c[n-2], c2[n-2]).
// Two kernels share the launch's waves (round 5): a wave whose 64 reads have (nearly) one length belongs to k_cumsum_gather, every other wave
// to k_cumsum -- each returns at once from the other's.
static __device__ __forceinline__ bool csum_uniform(int n) { return wave_max(n) - wave_min(n) < 8 * CK; }
__global__ void __launch_bounds__(64) k_cumsum(const float *__restrict__ down, const int32_t *__restrict__ nvalid, int Lp, int n_reads,
                                               int nck, double2 *__restrict__ ck, double2 *__restrict__ tail, int split)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    const int n = r < n_reads ? nvalid[r] : 0;
    if (split && csum_uniform(n)) return; // (k_cumsum_gather's wave; split = 0, ADP_CUMSUM_GATHER=0: every wave is this kernel's)
    if (r >= n_reads) return;
    const float *s = down + (size_t)r * Lp;
    double2 *c = ck + (size_t)r * nck;
    double a = 0.0, b = 0.0;
    double2 t = make_double2(0.0, 0.0);
    // one checkpoint block (CK = 16 pooled samples) per step; its four float4 loads are issued ahead of the add
    // chain (register sets used in turn, no moves): a lane reads its own row, so one wave-instruction touches 64 rows
    // and only the bytes in flight per wave buy bandwidth (one block ahead: 4.1 ms per 96 000 reads, waiting on memory at
    // every step).  The loads are unconditional -- rows are padded to Lp, the block index is clamped to the row -- because a
    // load behind a condition makes the compiler wait for it where the two paths join.
    const float4 *s4 = reinterpret_cast<const float4 *>(s); // rows start 256-byte aligned (Lp % 64 == 0)
    auto block = [&](int j0, const float4 (&q)[4]) {
        c[j0 / CK] = make_double2(a, b);
        const float v[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
        if (j0 + CK < n) { // a block in front of the read's last sample: no tests (convert, add, multiply, add per sample)
#pragma unroll
            for (int u = 0; u < CK; u++) { const double x = (double)v[u]; a += x; b += x * x; }
        } else {
#pragma unroll
            for (int u = 0; u < CK; u++) {
                const int j = j0 + u;
                if (j < n) {
                    if (j == n - 1) t = make_double2(a, b);
                    double x = (double)v[u];
                    a += x;
                    b += x * x;
                }
            }
        }
    };
    // A lane's loads come as whole 128-byte lines (two checkpoint blocks), PFL lines ahead (round 5; 64-byte blocks, three in flight: 3.15
    // against 2.95 ms per 96 000 reads; one block ahead: 4.1)
    constexpr int PFL = 2;
    float4 q[PFL][2][4];
    const int last_line = Lp / (2 * CK) - 1;
    auto ldl = [&](int line, float4 (&qq)[2][4]) {
        const float4 *p4 = s4 + (line < last_line ? line : last_line) * (2 * CK / 4);
#pragma unroll
        for (int e = 0; e < 8; e++) qq[e >> 2][e & 3] = p4[e];
    };
#pragma unroll
    for (int d = 0; d < PFL; d++) ldl(d, q[d]);
    for (int j0 = 0; j0 < n; j0 += PFL * 2 * CK) {
        const int line = j0 / (2 * CK);
#pragma unroll
        for (int d = 0; d < PFL; d++) {
            const int j1 = j0 + d * 2 * CK;
            if (d == 0 || j1 < n) block(j1, q[d][0]);
            if (j1 + CK < n) block(j1 + CK, q[d][1]);
            ldl(line + PFL + d, q[d]);
        }
    }
    tail[r] = t;
}

// k_cumsum for waves of one length.  What bounds k_cumsum is not its chains (21 cycles per sample: 0.4 ms per 96 000 reads,
// tools/chain_latency.hip) and not its loads: WITHOUT its checkpoint stores it takes 1.3 ms instead of 2.9 -- a 16-byte piece per lane and
// instruction into 64 different lines, every 16 samples (profiles/r05_tried_and_dropped.txt item 12).  Here the checkpoints wait in LDS for
// eight blocks and leave as whole 128-byte lines, eight reads per instruction; the loop is the wave's (all lanes to the longest read), which
// is why reads of very different lengths stay with k_cumsum.  Same additions in the same order.
#define CSUM_Q 8 // checkpoints gathered per read before they are stored: 8 x 16 bytes = one line
typedef double csum_d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k_cumsum_gather(const float *__restrict__ down, const int32_t *__restrict__ nvalid, int Lp, int n_reads,
                                                      int nck, double2 *__restrict__ ck, double2 *__restrict__ tail)
{
    __shared__ __attribute__((aligned(16))) csum_d2 cks_[64 * (CSUM_Q + 1)]; // [lane][CSUM_Q] with a piece of padding per row
    LDS csum_d2 *cks = (LDS csum_d2 *)cks_;
    const int ln = threadIdx.x, r0 = blockIdx.x * 64;
    const int r = r0 + ln;
    const int n = r < n_reads ? nvalid[r] : 0;
    if (!csum_uniform(n)) return; // (k_cumsum's wave; a uniform wave is a full one: r < n_reads for every lane, or all lengths below 8 CK)
    const int nmax = wave_max(n);
    const float *s = down + (size_t)(r < n_reads ? r : n_reads - 1) * Lp;
    double a = 0.0, b = 0.0;
    double2 t = make_double2(0.0, 0.0);
    const float4 *s4 = reinterpret_cast<const float4 *>(s); // rows start 256-byte aligned (Lp % 64 == 0)
    auto block = [&](int j0, int slot, const float4 (&q)[4]) {
        cks[ln * (CSUM_Q + 1) + slot] = (csum_d2){a, b};
        const float v[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
        if (j0 + CK < n) { // a block in front of the read's last sample: no tests (convert, add, multiply, add per sample)
#pragma unroll
            for (int u = 0; u < CK; u++) { const double x = (double)v[u]; a += x; b += x * x; }
        } else {
#pragma unroll
            for (int u = 0; u < CK; u++) {
                const int j = j0 + u;
                if (j < n) {
                    if (j == n - 1) t = make_double2(a, b);
                    double x = (double)v[u];
                    a += x;
                    b += x * x;
                }
            }
        }
    };
    static_assert(CSUM_Q % 4 == 0, "a group is a whole number of prefetch rounds of two lines");
    constexpr int PFL = 2, LPG = CSUM_Q / 2; // lines of two blocks per group of CSUM_Q blocks
    float4 q[PFL][2][4];
    const int last_line = n > 0 ? (n - 1) / (2 * CK) : 0; // (a read that has ended keeps asking for its last line: cached)
    auto ldl = [&](int line, float4 (&qq)[2][4]) {
        const float4 *p4 = s4 + (line < last_line ? line : last_line) * (2 * CK / 4);
#pragma unroll
        for (int e = 0; e < 8; e++) qq[e >> 2][e & 3] = p4[e];
    };
#pragma unroll
    for (int d = 0; d < PFL; d++) ldl(d, q[d]);
    for (int g0 = 0; g0 * (CSUM_Q * CK) < nmax; g0++) { // groups of CSUM_Q blocks = 128 samples, all lanes together (the stores below are the wave's)
#pragma unroll
        for (int l2 = 0; l2 < LPG; l2 += PFL) {
#pragma unroll
            for (int d = 0; d < PFL; d++) {
                const int line = g0 * LPG + l2 + d, j1 = line * 2 * CK, slot = 2 * (l2 + d);
                if (j1 < n) block(j1, slot, q[d][0]);
                if (j1 + CK < n) block(j1 + CK, slot + 1, q[d][1]);
                ldl(line + PFL, q[d]);
            }
        }
        // the group's checkpoints: lane l stores piece l & 7 of read 8 it + (l >> 3) -- eight reads' lines per instruction; a read's last line
        // is cut at its length (what lies at or beyond a read's last sample was never a checkpoint)
        ws_sync();
        const int q0 = g0 * CSUM_Q;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int rd = 8 * it + (ln >> 3), pc = ln & 7;
            const int nrd = __shfl(n, rd);
            if (r0 + rd < n_reads && (q0 + pc) * CK < nrd) {
                const csum_d2 v = cks[rd * (CSUM_Q + 1) + pc];
                ck[(size_t)(r0 + rd) * nck + q0 + pc] = make_double2(v.x, v.y);
            }
        }
        ws_sync();
    }
    if (r < n_reads) tail[r] = t;
}

// ---------------------------------------------------------------- gains
// The two quotients of var_c share their divisor (the segment length), so the refined reciprocal of the IEEE
// division sequence is computed once: y = v_rcp_f64(b) + two Newton steps, then per numerator
// q0 = a y, r = a - b q0 (exact, one fma), q = q0 + r y.  This is instruction for instruction what the compiler
// emits for a / b (v_div_scale / v_div_fmas / v_div_fixup only rescale operands near the ends of the exponent range
// and patch zero / infinite / NaN operands), so the quotients are the same bits as long as no scaling is needed:
// b is an integer in [1, 2^31), and the numerators are differences of running sums of float32 values or of their
// squares -- zero, or multiples of 2^-298 below 2^83.  An infinite or NaN sum ends as a NaN variance either way.
static __device__ __forceinline__ double recip_refined(double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-b, y1, 1.0);
    return __builtin_fma(y1, e1, y1);
}
static __device__ __forceinline__ double div_by_len(double a, double b, double y)
{
    const double q0 = a * y;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, y, q0);
}

// the library log for zero / negative / subnormal / non-finite arguments: out of line, it is next to never called
static __device__ __noinline__ double gains_log_slow(double u) { return log(u); }
static __device__ __forceinline__ double var_seg(double c2hi, double c2lo, double chi, double clo, double dl) // dl = (double)length
{
    const double y = recip_refined(dl);
    double mu = div_by_len(chi - clo, dl, y);
    return div_by_len(c2hi - c2lo, dl, y) - mu * mu;
}

// lane l takes lane l - 1's value, lane 0 takes `fill` (wave-uniform); all 64 lanes must be active
static __device__ __forceinline__ double wave_shr1_f64(double x, double fill)
{
    int2 v = __builtin_bit_cast(int2, x), f = __builtin_bit_cast(int2, fill);
    v.x = __builtin_amdgcn_update_dpp(f.x, v.x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    v.y = __builtin_amdgcn_update_dpp(f.y, v.y, 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, v);
}
static __device__ __forceinline__ double readlane_f64(double x, int lane)
{
    int2 v = __builtin_bit_cast(int2, x);
    v.x = __builtin_amdgcn_readlane(v.x, lane);
    v.y = __builtin_amdgcn_readlane(v.y, lane);
    return __builtin_bit_cast(double, v);
}

// grid = n_reads waves (block = 64).  PASS 1: start = 0, offsets (5, 5), also emits T1 (first / last
// index with a positive-or-NaN gain).  PASS 2: start = adapter candidate, offsets (1, 1).
// Emits the trace (float64) and per-64-point summaries: PASS 1 of the raw trace (NaN => +inf max),
// PASS 2 of the np.nan_to_num-sanitised trace that find_peaks sees in P4.
#ifndef GAINS_UNROLL
#define GAINS_UNROLL 1
#endif
#ifndef GAINS_SU
#define GAINS_SU 4
#endif
#ifndef GAINS_EU
#define GAINS_EU 4
#endif
#ifndef GAINS_WPB
#define GAINS_WPB 4 // waves (reads) per block
#endif
template <int PASS>
__global__ void __launch_bounds__(64 * GAINS_WPB, 4) k_gains(const float *__restrict__ down, const int32_t *__restrict__ nvalid, int Lp, int nck,
                                              const double2 *__restrict__ ck, const double2 *__restrict__ tail,
                                              const int32_t *__restrict__ adapter_idx, int mbsize,
                                              const MbState *__restrict__ mbs, double *__restrict__ trace,
                                              double *__restrict__ bmax, double *__restrict__ bmin, int nsum,
                                              int2 *__restrict__ t1, int sanitize, int32_t *__restrict__ pk_all,
                                              int32_t *__restrict__ npk_all, int pk_stride, double *__restrict__ gstat, int n_reads, int oh1,
                                              double *__restrict__ pkv_all = nullptr)
{
    // GAINS_WPB waves (reads) per block share log_cr's table; everything else is private to a wave, so the only
    // block-wide barrier is the one after the table copy
    __shared__ __attribute__((aligned(16))) double sg_[GAINS_WPB][64 * (CK + 1)];
    __shared__ __attribute__((aligned(16))) double lt_[3 * LOGCR_N]; // log_cr's table
    LDS double *sg = (LDS double *)sg_[threadIdx.x >> 6];
    const int r = blockIdx.x * GAINS_WPB + (threadIdx.x >> 6);
    const int ln = lane_id();
    for (int i = threadIdx.x; i < 3 * LOGCR_N; i += 64 * GAINS_WPB) lt_[i] = g_logcr_table[i];
    __syncthreads();
    if (r >= n_reads) return;
    const LDS double *lt = (const LDS double *)lt_;
    auto flog = [&](double v) { return log_cr_impl(v, lt, [](double u) { return gains_log_slow(u); }); };
    if (mbs[r / mbsize].status != ADP_MB_OK) return;
    const int n = nvalid[r];
    if (n <= 0) return;
    int start = 0, oh = oh1, ot = 5; // (oh1: 5 for combined_detect_llr2; 5 + min_obs_adapter // ds for the single-read API)
    if (PASS == 2) {
        start = adapter_idx[r];
        if (start < 0) return;
        oh = 1; ot = 1;
    }
    const int E = n - 1;
    const float *s = down + (size_t)r * Lp;
    const double2 *c = ck + (size_t)r * nck;
    double *g = trace + (size_t)r * Lp;
    // c[E-1], c2[E-1]
    const double2 te = tail[r];
    // c[start-1], c2[start-1] (sequential restart from the checkpoint, identical operations)
    double cs = 0.0, c2s = 0.0;
    if (PASS == 2 && start > 0) {
        int q = start / CK;
        double2 p = c[q];
        double a = p.x, b = p.y;
        for (int j = q * CK; j < start; j++) { double v = (double)s[j]; a += v; b += v * v; }
        cs = a; c2s = b;
    }
    double vs;
    {
        double v = (start == E) ? 0.0 : var_seg(te.y, c2s, te.x, cs, (double)(E - start));
        vs = (double)(E - start) * flog(v);
    }
    int first_pos = 0x7fffffff, last_pos = -1;
    double st_s1 = 0.0, st_s2 = 0.0; // PASS 1: sum and sum of squares of the non-NaN trace values (for np.nanstd in P1)
    int st_nan = 0;
    // P4 needs every strict local maximum of the sanitised trace: they are picked up here, while the tile
    // is in LDS, and handed to k_polya_peak as an index list (npk = -1: a plateau was met, recount there)
    const bool emit = (PASS == 2 || sanitize) && pk_all;
    int32_t *pk = emit ? pk_all + (size_t)r * pk_stride : nullptr;
    // the maxima's (sanitised) heights beside their positions: k_polya_peak reads them as a list instead of gathering one cache line per maximum
    double *pkv = (emit && pkv_all) ? pkv_all + (size_t)r * pk_stride : nullptr;
    int npk = 0;
    bool plateau = false;
    double carry1 = 0.0, carry2 = 0.0; // sanitised x[tb-1], x[tb-2]
    for (int tb = 0; tb < n; tb += TRACE_TILE) {
        // each lane walks CK = 16 consecutive pooled samples: four float4 loads (rows start 256-byte aligned and
        // are padded to Lp, so the loads stay inside the row; samples at or beyond n are not used)
        ws_sync();
        const int i0 = tb + ln * CK;
        const float4 *s4 = reinterpret_cast<const float4 *>(s + (i0 < Lp ? i0 : 0));
        double a = 0.0, b = 0.0;
        if (i0 < n) { double2 p = c[i0 / CK]; a = p.x; b = p.y; }
        double mx = -__builtin_inf(), mn = __builtin_inf();
        // INTERIOR tiles -- every point has both segments and lies inside the read -- take a body without the range
        // tests, and the rare values (NaN, infinities) leave the straight path through branches: the kernel is bound
        // by float64 instruction issue, and scalar branches cost no vector slots where selects do.
        // the two segment lengths of a lane's points as float64 counters (+1 / -1 per point, exact): an int -> float64
        // conversion runs at a quarter of the add rate, and the kernel is bound by float64 issue (profiles/r02_sq_counters_*)
        double dlh = (double)(i0 - start), dlt = (double)(E - i0);
        auto point = [&](auto interior_c, int t, int i, float qf) {
            constexpr bool INTERIOR = decltype(interior_c)::value;
            double gi = 0.0;
            const double lh_len = dlh, lt_len = dlt;
            dlh += 1.0; dlt -= 1.0;
            if (INTERIOR || i < n) {
                if (INTERIOR || (i >= start + oh && i < E - ot)) {
                    double vh = var_seg(b, c2s, a, cs, lh_len);
                    double vt = var_seg(te.y, b, te.x, a, lt_len);
                    // both logarithms side by side (independent instruction streams), exceptions patched afterwards
#ifdef ADP_GAINS_LOG_CR
                    double lh = log_cr_fast(vh, lt), ll = log_cr_fast(vt, lt);
#else
                    double lh = log_1ulp_fast(vh, lt), ll = log_1ulp_fast(vt, lt); // (< 1 ULP in 20 operations: log_cr.h)
#endif
                    if (!(log_cr_ok(vh) && log_cr_ok(vt))) {
                        if (!log_cr_ok(vh)) lh = gains_log_slow(vh);
                        if (!log_cr_ok(vt)) ll = gains_log_slow(vt);
                    }
                    double h = lh_len * lh;
                    double tl = lt_len * ll;
                    gi = vs - (h + tl);
                }
                if (PASS == 1 && !(gi <= 0.0)) { first_pos = min(first_pos, i); last_pos = i; } // (i grows along a lane)
                const bool finite = __builtin_fabs(gi) < __builtin_inf();
                if (PASS == 1) { if (gi == gi) { st_s1 += gi; st_s2 += gi * gi; } else st_nan++; }
                if (PASS == 1 && !sanitize) {
                    if (finite) { mx = gi > mx ? gi : mx; mn = gi < mn ? gi : mn; }
                    else {
                        double m1 = (gi != gi) ? __builtin_inf() : gi;
                        mx = m1 > mx ? m1 : mx;
                        if (gi == gi) mn = gi < mn ? gi : mn;
                    }
                } else {
                    double x = gi;
                    if (!finite) x = (x != x) ? 0.0 : (x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308);
                    mx = x > mx ? x : mx;
                    mn = x < mn ? x : mn;
                }
                double v = (double)qf;
                a += v;
                b += v * v;
            }
            sg[ln * (CK + 1) + t] = gi;
        };
        const bool interior = tb >= start + oh && tb + TRACE_TILE <= E - ot; // (then also tb + TRACE_TILE <= n)
        if (interior) {
#pragma unroll GAINS_UNROLL
            for (int tq = 0; tq < CK / 4; tq++) {
                const float4 q4 = s4[tq];
                const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int u = 0; u < 4; u++) point(std::true_type{}, tq * 4 + u, i0 + tq * 4 + u, qv[u]);
            }
        } else {
#pragma unroll GAINS_UNROLL
            for (int tq = 0; tq < CK / 4; tq++) {
                const float4 q4 = s4[tq];
                const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int u = 0; u < 4; u++) point(std::false_type{}, tq * 4 + u, i0 + tq * 4 + u, qv[u]);
            }
        }
        // 4 lanes = one 64-point summary block
        {
            double w;
            w = __shfl_xor(mx, 1); mx = w > mx ? w : mx;
            w = __shfl_xor(mx, 2); mx = w > mx ? w : mx;
            w = __shfl_xor(mn, 1); mn = w < mn ? w : mn;
            w = __shfl_xor(mn, 2); mn = w < mn ? w : mn;
            int blk = tb / SUMBLK + (ln >> 2);
            if ((ln & 3) == 0 && blk < nsum && (tb + (ln >> 2) * SUMBLK) < n) {
                bmax[(size_t)r * nsum + blk] = mx;
                bmin[(size_t)r * nsum + blk] = mn;
            }
        }
        ws_sync();
        auto sanit = [](double x) { // np.nan_to_num
            if (x != x) x = 0.0;
            else if (__builtin_isinf(x)) x = x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
            return x;
        };
        auto sval = [&](int e) { return sanit(sg[(e / CK) * (CK + 1) + (e % CK)]); }; // sanitised value at tile offset e (e >= 0)
        if (!emit) {
#pragma unroll GAINS_SU
            for (int k = 0; k < CK; k++) {
                int e = k * 64 + ln;
                int i = tb + e;
                if (i < n) g[i] = sg[(e / CK) * (CK + 1) + (e % CK)];
            }
        } else {
            // ONE read of the staging tile serves both the trace's write-out and the pick-up of local maxima: the element a lane
            // stores is its "next" value, the two in front of it are its neighbours' -- a wave shift (v_mov_dpp wave_shr:1, lane 0
            // filled with what lane 63 / 62 held in the round before, or the tile's carries) instead of two more sanitised LDS reads
            // with their index arithmetic per point (the pick-up was ~200 of pass 2's ~985 cycles per wave-point).
            double p1 = carry1, p2 = carry2; // x[e - 1], x[e - 2] of lane 0 (wave-uniform)
#pragma unroll GAINS_EU
            for (int k = 0; k < CK; k++) {
                const int e = k * 64 + ln;
                const int i = tb + e;   // x[i] is "next"; the candidate is j = i - 1
                const int j = i - 1;
                const double raw = sg[(e / CK) * (CK + 1) + (e % CK)];
                if (i < n) g[i] = raw;
                const double xn = sanit(raw);
                const double xj = wave_shr1_f64(xn, p1);
                const double xp = wave_shr1_f64(xj, p2);
                p1 = readlane_f64(xn, 63); p2 = readlane_f64(xn, 62);
                bool pkf = false;
                if (i < n && j >= 1) {
                    if (xp < xj) {
                        if (xn < xj) pkf = true;
                        else if (xn == xj && !(xj == 0.0 && j >= E - ot)) plateau = true; // trailing zeros never peak
                    }
                }
                unsigned long long mk = __ballot(pkf);
                if (pkf) { const int at = npk + __popcll(mk & ((1ull << ln) - 1ull)); pk[at] = j; if (pkv) pkv[at] = xj; }
                npk += __popcll(mk);
            }
            const int last = min(TRACE_TILE, n - tb) - 1;
            double c1 = sval(last), c2v = (last >= 1) ? sval(last - 1) : carry1;
            carry2 = c2v; carry1 = c1;
        }
    }
    if (emit) {
        plateau = __any(plateau);
        if (ln == 0) npk_all[r] = plateau ? -1 : npk;
    }
    if (PASS == 1) {
        first_pos = wave_min(first_pos);
        last_pos = wave_max(last_pos);
        if (ln == 0) {
            // np.argmin(signal <= 0): first False, 0 if none; end = size - argmin(reversed) - 1
            int st = (last_pos < 0) ? 0 : first_pos;
            int en = (last_pos < 0) ? n - 1 : last_pos;
            t1[r] = make_int2(st, en);
        }
        st_s1 = wave_sum(st_s1); st_s2 = wave_sum(st_s2); st_nan = wave_sum(st_nan);
        if (ln == 0 && gstat) { gstat[3 * r] = st_s1; gstat[3 * r + 1] = st_s2; gstat[3 * r + 2] = (double)st_nan; }
    }
}
