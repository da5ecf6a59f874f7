// cnn_conv_split.h -- C2 on the float16 matrix cores at float32 accuracy (the default conv stack; cnn_conv.h is the exact-float32 one).
//
// reference adapted/detect/cnn.py:16-52 (BoundariesCNN), :85-98 (cnn_score): float32 throughout.  gfx950 has no fast float32
// matrix path (v_mfma_f32_32x32x2_f32 runs at the vector rate, 1/16 of the float16 rate), so the two 64 -> 64 layers -- 98 % of the
// arithmetic -- are computed from SPLIT operands instead: every float32 value v is carried as two float16 numbers
//       hi = RN16(v),   lo = RN16((v - hi) * 2^11)            v = hi + lo 2^-11 (1 + e),  |e| <= 2^-11  ->  22 significant bits
// (the residual is scaled so that it stays a normal float16 whatever v's magnitude), and a product of two such values is
//       a w = a.hi w.hi + (a.hi w.lo + a.lo w.hi) 2^-11 + O(2^-22 |a w|)
// three v_mfma_f32_32x32x16_f16 per 16 k-values, float16 products exact in the float32 accumulators, the cross terms in an
// accumulator of their own that is folded in once per tile.  Error per product <= 3 * 2^-22 = 7e-7 relative (random in sign),
// the same order as the rounding of a float32 fmaf chain over K = 448 terms; measured against torch float32 in
// tests/test_gpu_cnn.py beside the exact-float32 kernels.  3 MFMAs x 32 cycles per 32 x 32 x 16 block against 8 x 64 for the float32
// instruction: 5.3 x the matrix rate -- the layers become HBM-bound (1.8 MB read + 1.8 MB written per read and layer).
//
// Values outside the float16 range: activations are stored times 2^-4 (real values below 5.2e5 asked for: a 1500-pA artefact in a
// read normalised by a MAD of 10 stays far inside); a kernel that produces a larger or non-finite activation anywhere raises the
// CALL's flag (a word beside the open-pore arena counter, read with it at the end of the
// call: no extra host round trip), and the call is repeated on the float32 kernels of cnn_conv.h -- inputs like 1e30 pA or infinities
// give the rows the float32 stack gives.  Weights are scaled per layer by a power of two (largest |w| s in [2^13, 2^14)); the factor
// leaves with the bias in the epilogue.
//
// Layout: activations in HBM are CHANNEL-LAST rows, one per position: 64 hi | 64 lo | 8 pad float16 = 272 bytes, CNS_FRONT zero
// rows in front of position 0 and zero rows behind L1 (the "same" padding is read, not branched on).  A workgroup step covers
// PB = 64 NT positions x 64 channels: its input tile (PB + 6 rows) is ONE contiguous byte range, copied by LDS-DMA into the second
// LDS buffer while the first is computed on.  The 272-byte row pitch shifts consecutive rows by 4 banks, so the A fragment of a
// k-step -- one ds_read_b128 per lane and plane: 8 consecutive channels of one position and tap -- is conflict-free.
// GEMM: M = output channels (A = weights, 2 planes x 28 k-steps x 4 VGPRs resident in registers), N = positions (B = activations
// from LDS), K = (tap, channel).
#pragma once
#include "cnn_conv.h"
#ifdef ADP_PHASE_TIMING
#include "common.h"
extern __device__ unsigned long long g_dbg[ADP_NDBG];
#endif

typedef _Float16 cnn_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cnn_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 cnn_h2 __attribute__((ext_vector_type(2)));
typedef float cnn_f2 __attribute__((ext_vector_type(2)));
typedef float cnn_f4 __attribute__((ext_vector_type(4)));
typedef unsigned short cnn_us2 __attribute__((ext_vector_type(2)));
#ifndef CNS_ABL
#define CNS_ABL 0 // (development: 1 = no MFMAs, 2 = no epilogue stores)
#endif

template <int N, class F, int I = 0>
static __device__ __forceinline__ void cns_static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); cns_static_for<N, F, I + 1>(static_cast<F &&>(f)); }
}

#define CNS_ROW 136          // float16 per position row
#define CNS_ROWB 272         // bytes
#define CNS_FRONT 4          // zero rows in front of position 0
#define CNS_KSTEPS 28        // 7 taps x 4 groups of 16 channels
#define CNS_LIMIT 32768.0f   // a STORED activation at or beyond it (or non-finite) repeats the call on the float32 kernels ...
#define CNS_ASCALE 0.0625f   // ... and activations are stored times 2^-4: real values up to 5.2e5 stay in range (exact scaling; it
                             // rides through the 64 -> 64 layers with the bias and leaves in layer 3)
#define CNS_SLACK 131072     // bytes behind the last read's rows that a tile DMA may read (k_cnn_conv_out_s: up to 130 rows from a row below L1;
                             // the overlapping tiles of the folded last layer: up to 64 NT rows)
#define CNS_WSP_LAYER (2 * CNS_KSTEPS * 2 * 64 * 8) // float16 per layer in the split-weight buffer
// Row 0 of a read's front padding is the DUMP row: the epilogues store the pieces of rows at or beyond L1 there so that every store is issued
// and the counted s_waitcnt stays exact.  Nothing may read it: a tile's DMA starts at row CNS_FRONT - 3, layer 3's block at CNS_FRONT - 1.
static_assert(CNS_FRONT >= 4, "the dump row (row 0) must lie in front of every row a tile DMA or the last layer reads");

static __device__ __forceinline__ void cns_split(float y, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)y;
    lo = (_Float16)((y - (float)hi) * 2048.0f);
}
static __device__ __forceinline__ float cns_join(_Float16 hi, _Float16 lo) { return __builtin_fmaf((float)lo, 1.0f / 2048.0f, (float)hi); }

// ---------------------------------------------------------------- weights of a 64 -> 64 layer as B fragments
// wsp [mh 2][k-step 28][plane 2][lane 64][8]: lane l of channel half mh holds W[o = 32 mh + (l & 31)][c = 16 cg + 8 (l >> 5) + e][t],
// k-step = 4 t + cg, scaled by sw and split.
__global__ void k_cns_split_weights(const float *__restrict__ w /* [64][64][7] */, float sw, _Float16 *__restrict__ wsp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * CNS_KSTEPS * 64) return;
    const int lane = idx & 63, k = (idx >> 6) % CNS_KSTEPS, mh = idx / (64 * CNS_KSTEPS);
    const int t = k >> 2, cg = k & 3, o = 32 * mh + (lane & 31), c0 = 16 * cg + 8 * (lane >> 5);
    _Float16 *dh = wsp + ((size_t)((mh * CNS_KSTEPS + k) * 2 + 0) * 64 + lane) * 8;
    _Float16 *dl = wsp + ((size_t)((mh * CNS_KSTEPS + k) * 2 + 1) * 64 + lane) * 8;
    for (int e = 0; e < 8; e++) {
        const float v = w[((size_t)o * CNN_C + c0 + e) * CNN_K + t] * sw;
        _Float16 hi, lo;
        cns_split(v, hi, lo);
        dh[e] = hi; dl[e] = lo;
    }
}

// ---------------------------------------------------------------- weights of layer 3 (ConvTranspose1d 64 -> 2, k 7) as A fragments
// For the fold of layer 3 into layer 2's kernel: one more GEMM per tile with M = 16 rows (of the instruction's 32), K = the 64 channels,
// N = positions.  Row m stands for (output channel o = (m >> 2) & 1, tap t = (m & 3) + 4 (m >> 3)), t = 7 and the rows from 16 on are
// zero: in the accumulator layout of v_mfma_f32_32x32x16 (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) lane half `lane >> 5` then holds
// output channel o = lane >> 5 and register t the partial sum of tap t -- the seven taps of one position and channel in registers 0..6.
// w3sp [k-step 4][plane 2][lane 64][8]: lane l holds W3[c = 16 cg + 8 (l >> 5) + e][o][t] for row m = l & 31, scaled by s3 and split.
#define CNS_W3SP (4 * 2 * 64 * 8) // float16
__global__ void k_cns_split_w3(const float *__restrict__ w3 /* [64][2][7] */, float s3, _Float16 *__restrict__ w3sp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * 64) return;
    const int lane = idx & 63, cg = idx >> 6;
    const int m = lane & 31, o = (m >> 2) & 1, t = (m & 3) + 4 * (m >> 3), c0 = 16 * cg + 8 * (lane >> 5);
    _Float16 *dh = w3sp + ((size_t)(cg * 2 + 0) * 64 + lane) * 8, *dl = w3sp + ((size_t)(cg * 2 + 1) * 64 + lane) * 8;
    for (int e = 0; e < 8; e++) {
        const float v = (m < 16 && t < CNN_K) ? w3[((size_t)(c0 + e) * 2 + o) * CNN_K + t] * s3 : 0.0f;
        _Float16 hi, lo;
        cns_split(v, hi, lo);
        dh[e] = hi; dl[e] = lo;
    }
}

// ---------------------------------------------------------------- weights of layer 0 (Conv1d 1 -> 64, k 7, stride 3) as A fragments
// For layer 0 INSIDE layer 1's kernel (FIRST): the tile's input rows are one small GEMM per 32 rows -- M = 32 channels of the wave's half,
// K = the 7 taps padded to 16 (the second k-half is zero), N = rows; B = the samples x[3 p - 3 + tap], split like every other operand.
// w0sp [plane 2][mh 2][lane 64][8]: lane l holds W0[c = 32 mh + (l & 31)][tap e] for l < 32 and e < 7 (zeros elsewhere), times s0, split.
#define CNS_W0SP (2 * 2 * 64 * 8) // float16
__global__ void k_cns_split_w0(const float *__restrict__ w0 /* [64][1][7] */, float s0, _Float16 *__restrict__ w0sp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * 64) return;
    const int lane = idx & 63, mh = idx >> 6;
    const int c = 32 * mh + (lane & 31);
    _Float16 *dh = w0sp + ((size_t)(0 * 2 + mh) * 64 + lane) * 8, *dl = w0sp + ((size_t)(1 * 2 + mh) * 64 + lane) * 8;
    for (int e = 0; e < 8; e++) {
        const float v = (lane < 32 && e < CNN_K) ? w0[(size_t)c * CNN_K + e] * s0 : 0.0f;
        _Float16 hi, lo;
        cns_split(v, hi, lo);
        dh[e] = hi; dl[e] = lo;
    }
}
struct CnsFirst { const float *x; int Lc; const _Float16 *w0sp; const float *b0; float s0; }; // the prepared signal [n][Lc], layer 0's fragments, bias, scale

// ---------------------------------------------------------------- layer 0: Conv1d(1 -> 64, k 7, stride 3, pad 3) + ReLU, split rows out
// grid = (ceil(L1 / 64), n); block = 256: 64 positions, wave q makes channels 16 q .. 16 q + 15 of all of them (weights wave-uniform:
// scalar operands; the float32 fmaf chain of k_cnn_conv_in, two channels per v_pk_fma_f32).  The rows are put together in LDS
// (272-byte pitch: conflict-free 16-byte writes) and leave as one contiguous, fully coalesced copy -- written from the threads
// directly, a wave's store would touch 64 different rows.  17 KB of LDS and 256 threads per block: eight blocks per CU, the
// copies of some behind the arithmetic of others.
#define CNS_IN_P 64
#define CNS_IN_T 4 // position tiles per block (the weights are fetched once per block: as scalar operands they were a chain of
                   // dependent scalar-cache round trips in front of every tile's arithmetic)
__global__ void __launch_bounds__(256) k_cnn_conv_in_s(const float *__restrict__ x, int Lc, int L1, int Lrows,
                                                        const float *__restrict__ w /* [64][1][7] */, const float *__restrict__ b,
                                                        _Float16 *__restrict__ out, int32_t *__restrict__ flag)
{
    __shared__ __attribute__((aligned(16))) _Float16 rows_[CNS_IN_P * CNS_ROW];
    LDS _Float16 *rows = (LDS _Float16 *)rows_;
    const int n = blockIdx.y;
    const int pl = threadIdx.x & (CNS_IN_P - 1), qw = threadIdx.x >> 6;
    const float *row = x + (size_t)n * Lc;
    // the wave's 16 channels as 8 pairs: weights and biases in vector registers (every lane the same values, all loads in flight at once)
    cnn_f2 wp[8][CNN_K], bp[8];
    {
        const float *wq = w + 16 * qw * CNN_K, *bq = b + 16 * qw;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            bp[e] = (cnn_f2){bq[2 * e], bq[2 * e + 1]};
#pragma unroll
            for (int t = 0; t < CNN_K; t++) wp[e][t] = (cnn_f2){wq[(2 * e) * CNN_K + t], wq[(2 * e + 1) * CNN_K + t]};
        }
    }
    cnn_us2 hmax = {0, 0};
    LDS _Float16 *o = rows + pl * CNS_ROW + 16 * qw;
    for (int tile = 0; tile < CNS_IN_T; tile++) {
        const int p0 = (blockIdx.x * CNS_IN_T + tile) * CNS_IN_P;
        if (p0 >= L1) break;
        const int p = p0 + pl;
        float v[CNN_K];
#pragma unroll
        for (int t = 0; t < CNN_K; t++) { const int i = 3 * p + t - 3; v[t] = (i >= 0 && i < Lc) ? row[i] : 0.f; }
#pragma unroll
        for (int c8 = 0; c8 < 2; c8++) {
            cnn_h2 hq[4], lq[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                cnn_f2 acc = bp[c8 * 4 + e];
#pragma unroll
                for (int t = 0; t < CNN_K; t++) acc = __builtin_elementwise_fma(wp[c8 * 4 + e][t], (cnn_f2){v[t], v[t]}, acc);
                acc = __builtin_elementwise_max(acc, (cnn_f2){0.f, 0.f}) * CNS_ASCALE; // (a NaN becomes 0, as `acc > 0 ? acc : 0` makes it in k_cnn_conv_in)
                const cnn_h2 hi = __builtin_convertvector(acc, cnn_h2);
                const cnn_f2 rs = (acc - __builtin_convertvector(hi, cnn_f2)) * 2048.0f;
                hq[e] = hi; lq[e] = __builtin_convertvector(rs, cnn_h2);
                hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(cnn_us2, hi));
            }
            const cnn_h8 hh = {hq[0][0], hq[0][1], hq[1][0], hq[1][1], hq[2][0], hq[2][1], hq[3][0], hq[3][1]};
            const cnn_h8 ll = {lq[0][0], lq[0][1], lq[1][0], lq[1][1], lq[2][0], lq[2][1], lq[3][0], lq[3][1]};
            *reinterpret_cast<LDS cnn_h8 *>(o + c8 * 8) = hh;
            *reinterpret_cast<LDS cnn_h8 *>(o + 64 + c8 * 8) = ll;
        }
        __syncthreads();
        // 17 16-byte pieces per row (the last one is the row's padding: stays as it is), rows at or beyond L1 stay zero
        const int nrows = L1 - p0 < CNS_IN_P ? L1 - p0 : CNS_IN_P;
        cnn_h8 *dst = reinterpret_cast<cnn_h8 *>(out + ((size_t)n * Lrows + CNS_FRONT + p0) * CNS_ROW);
        const LDS cnn_h8 *src = reinterpret_cast<const LDS cnn_h8 *>(rows);
        for (int i = threadIdx.x; i < nrows * 17; i += 256)
            dst[i] = src[i]; // (the 17th piece of a row, its padding, goes along: a hole per row would make every line a partial write)
        __syncthreads();
    }
    // out of range: a hi part of 32768 or more, or infinite (the values are non-negative: bit patterns order like values); rows at
    // or beyond L1 see zeros or real samples like their neighbours -- a needless flag only costs the float32 repeat
    const bool bad = hmax[0] >= 0x7800 || hmax[1] >= 0x7800;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------- layers 1, 2: Conv1d(64 -> 64, k 7, pad 3) + ReLU, split float16 MFMA
// grid = persistent (one workgroup per CU); block = 256 = 4 waves = 2 channel halves (mh) x 2 position halves (ph);
// dynamic LDS = 2 tiles of (64 NT + 6) rows, each rounded up to whole 1 KB DMA instructions.
// GEMM orientation: M = output channels (A = weights), N = positions (B = activations), so D has the POSITION on the lane and four
// consecutive channels in four consecutive registers: a lane stores 8 bytes (4 float16) per plane and channel group.
// Synchronisation per step: the tile's DMA is the OLDEST of the wave's outstanding vector-memory operations (it was issued before
// the previous step's stores), so a counted s_waitcnt leaves those stores in flight; every store of the epilogue is issued
// unconditionally (rows at or beyond L1 go to the read's row 0, which nothing reads) to keep that count exact.
// LAST (round 5): layer 3 -- ConvTranspose1d(64 -> 2, k 7, stride 3, pad 3) -- in this kernel's epilogue.  The tile's output rows are
// laid down in LDS as before (with the rows outside [0, L1) zeroed: layer 3 reads them as its padding), but instead of leaving for HBM
// they are the B operand of one more GEMM (k_cns_split_w3's A fragments: 12 MFMAs per 32 positions); the partial sums of a position's
// seven taps go to LDS, and output 3 j + d of channel o is b + P[j - 1][6] + P[j][3] + P[j + 1][0] (d = 0), P[j][3 + d] + P[j + 1][d]
// (d = 1, 2).  A tile's first and last position only lend their partial sums: tiles advance by PB - 2 positions and start at -1.
// No rows of layer 2 in HBM (1.8 MB per read written and read back) and no k_cnn_conv_out_s.
// FIRST (round 5): layer 0 -- Conv1d(1 -> 64, k 7, stride 3, pad 3) + ReLU -- in this kernel's prologue.  No input rows in HBM: the
// tile's input rows are made in LDS from the prepared signal by 3 MFMAs per 32 rows and channel half (k_cns_split_w0's A fragments;
// the samples of a row are its B fragment, loaded a tile ahead), converted like an epilogue's results; rows outside [0, L1) are the
// zero padding.  Tiles advance by PB - 6 positions: the PB rows made are what PB - 6 positions need.  (Round 3 made these rows with float32 fmaf chains in the one wave per SIMD: no gain; on the matrix cores they cost
// 11 MFMAs per wave and tile.)  No k_cnn_conv_in_s, 1.8 MB per read not written and not read back.
template <int NT, bool LAST = false, bool FIRST = false>
__global__ void __launch_bounds__(256, 1) k_cnn_conv64s(const _Float16 *__restrict__ in, _Float16 *__restrict__ out,
                                                        const _Float16 *__restrict__ wsp, const float *__restrict__ bias, float sw,
                                                        float inv_sw, int n_reads, int L1, int Lrows, int tiles_per_read,
                                                        int32_t *__restrict__ flag, const _Float16 *__restrict__ w3sp = nullptr,
                                                        const float *__restrict__ b3 = nullptr, float inv_s3 = 0.f,
                                                        float *__restrict__ scores = nullptr, int Lo = 0, CnsFirst F = CnsFirst{})
{
    static_assert(!(FIRST && LAST), "layer 1 takes layer 0 in, layer 2 takes layer 3 in");
    constexpr int PB = 64 * NT, R = PB + 6, TILE_B = (R * CNS_ROWB + 1023) / 1024 * 1024, NDMA = TILE_B / 1024;
    // positions a tile advances by.  FIRST: PB - 6, so that the rows a tile needs (PBS + 6) are exactly its 2 NT subtiles of 32 -- a seventh
    // subtile for six rows cost a quarter of the prologue at NT = 3; the tile's last six positions are computed from rows nobody made and
    // are dropped (zeroed before the range check, not stored): 3 % more tiles
    constexpr int PBS = LAST ? PB - 2 : FIRST ? PB - 6 : PB;
    constexpr int NSTORE = (PB * 17 + 255) / 256; // vector-memory instructions of one epilogue: 16-byte pieces of PB rows over 256 threads
    static_assert(NSTORE * 256 >= PB * 17 && (NSTORE - 1) * 256 < PB * 17, "the epilogue issues exactly NSTORE stores per thread: the s_waitcnt below counts them");
    extern __shared__ __attribute__((aligned(16))) float cns_lds_raw[];
    LDS char *lds = (LDS char *)cns_lds_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mh = wave & 1, ph = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // A fragments: lane l holds W[o = 32 mh + (l & 31)][c = 16 cg + 8 (l >> 5) + e][t] of k-step 4 t + cg, e = 0..7
    cnn_h8 wh[CNS_KSTEPS], wl[CNS_KSTEPS];
    {
        const cnn_h8 *wp = reinterpret_cast<const cnn_h8 *>(wsp) + (size_t)(mh * CNS_KSTEPS) * 2 * 64 + lane;
#pragma unroll
        for (int k = 0; k < CNS_KSTEPS; k++) { wh[k] = wp[(size_t)(k * 2) * 64]; wl[k] = wp[(size_t)(k * 2 + 1) * 64]; }
    }
    // D layout: column = lane & 31 (position), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel inside the wave's 32)
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bs[r] = bias[32 * mh + (r & 3) + 8 * (r >> 2) + 4 * lh] * (sw * CNS_ASCALE); // (inputs and outputs carry CNS_ASCALE)

    cnn_f32x16 bsv;
#pragma unroll
    for (int r = 0; r < 16; r++) bsv[r] = bs[r];
    // (FIRST at NT = 4 is over the register limit by about this vector: there it waits in LDS and is fetched per tile)
    constexpr bool BS_LDS = FIRST && NT == 4;
    const cnn_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int total = n_reads * tiles_per_read; // (the host keeps it below 2^31)
    auto dma = [&](int tix, int b) {
        const int n = tix / tiles_per_read, tile = tix - n * tiles_per_read;
        // rows (position tile * PB - 3) ... of read n: one contiguous range (the instructions of the last KB read a little past it,
        // into the next rows or the buffer's slack)
        const GLB char *src = (const GLB char *)in + ((size_t)n * Lrows + CNS_FRONT - 3 - (LAST ? 1 : 0) + (size_t)tile * PBS) * CNS_ROWB + lane * 16;
        for (int inst = wave; inst < NDMA; inst += 4)
            __builtin_amdgcn_global_load_lds((const GLB float *)(src + inst * 1024), (LDS float *)(lds + b * TILE_B + inst * 1024), 16, 0, 0);
    };
    // LAST: layer 3's A fragments (32 registers) stay resident where the tile shape leaves room (NT = 4 is at the limit: fetched per tile)
    float b3_0 = 0.f, b3_1 = 0.f;
    if constexpr (LAST) { b3_0 = b3[0]; b3_1 = b3[1]; }
    constexpr bool W3_RESIDENT = LAST && NT < 4;
    cnn_h8 ah[4], al[4];
    if constexpr (W3_RESIDENT) {
        const cnn_h8 *wp = reinterpret_cast<const cnn_h8 *>(w3sp) + lane;
#pragma unroll
        for (int cg = 0; cg < 4; cg++) { ah[cg] = wp[(size_t)(cg * 2) * 64]; al[cg] = wp[(size_t)(cg * 2 + 1) * 64]; }
    }
    // FIRST: the second tile buffer is not one (the rows are made in place): it holds the samples of the next tile's rows (LDS-DMA, a dword
    // per lane: x[3 (tile PBS - 3) - 3 ..], 3 PB + 4 of them), layer 0's A fragments and its bias vector -- nothing of layer 0 in registers
    // across the k-loop
    constexpr int NRS = 2 * NT, NXD = (3 * PB + 4 + 63) / 64; // (rows made per tile: PB = PBS + 6)
    constexpr int XOFF = TILE_B;
    constexpr int W0OFF = XOFF + 4096, B0OFF = XOFF + 8192;
    static_assert(NXD * 256 + 7 * 4 + 3 * 32 * 4 <= 4096 && B0OFF + 512 <= 2 * TILE_B, "layer 0's operands fit in the second tile buffer");
    float k1_ = 0.f, k2_ = 0.f;
    if constexpr (FIRST) {
        reinterpret_cast<LDS cnn_h8 *>(lds + W0OFF)[threadIdx.x] = reinterpret_cast<const cnn_h8 *>(F.w0sp)[threadIdx.x]; // [plane][mh][lane]
        if (threadIdx.x < CNN_C) reinterpret_cast<LDS float *>(lds + B0OFF)[threadIdx.x] = F.b0[threadIdx.x] * F.s0;
        else if (BS_LDS && threadIdx.x < 2 * CNN_C) reinterpret_cast<LDS float *>(lds + B0OFF)[threadIdx.x] = bias[threadIdx.x - CNN_C] * (sw * CNS_ASCALE);
        k1_ = CNS_ASCALE / F.s0; k2_ = k1_ * (1.0f / 2048.0f);
    }
    auto load_x = [&](int tix) { // (every piece is issued: an index outside the signal reads its nearest sample, and make_rows puts the zero in)
        const int n = tix / tiles_per_read, tile = tix - n * tiles_per_read;
        const float *xr = F.x + (size_t)n * F.Lc;
        const int base = 3 * (tile * PBS - 3) - 3;
        for (int inst = wave; inst < NXD; inst += 4) {
            int i = base + inst * 64 + lane;
            i = i < 0 ? 0 : i >= F.Lc ? F.Lc - 1 : i;
            __builtin_amdgcn_global_load_lds((const GLB float *)(xr + i), (LDS float *)(lds + XOFF + inst * 256), 4, 0, 0);
        }
    };
    int it = blockIdx.x;
    int buf = 0;
    // out of range = a float16 hi part of 32768 or more (or infinite), looked for as the largest bit pattern seen: the values are
    // non-negative behind the ReLU, so patterns order like values.  (Rows at or beyond L1 of a read's last tile take part; they
    // hold what real neighbours hold, and a needless flag only costs the float32 repeat.)
    cnn_us2 hmax = {0, 0};
    const float cx = inv_sw * (1.0f / 2048.0f);
    // FIRST: the rows of tile tix -- positions tile PB - 3 .. -- from the samples load_x(tix) brought, into tile buffer 0: wave (mh, ph)
    // makes channels 32 mh .. of the row subtiles ph, ph + 2, ...; B fragment: lane = (row 32 rs + l31, k-half lh), the second k-half zeros
    auto make_rows = [&](int tix) {
        const int n = tix / tiles_per_read, tile = tix - n * tiles_per_read;
        (void)n;
        // (the lane index made opaque per call: the compiler otherwise keeps every subtile's row, sample and store address in registers
        // across the k-loop, which at NT = 4 has none to spare -- they went to scratch, and a scratch reload waits for the previous
        // step's stores)
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));
        const int l31 = ln_ & 31, lh = ln_ >> 5;
        const cnn_h8 w0h = reinterpret_cast<const LDS cnn_h8 *>(lds + W0OFF)[(0 * 2 + mh) * 64 + lane];
        const cnn_h8 w0l = reinterpret_cast<const LDS cnn_h8 *>(lds + W0OFF)[(1 * 2 + mh) * 64 + lane];
        cnn_f32x16 bs0v;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const cnn_f4 b4 = *reinterpret_cast<const LDS cnn_f4 *>(lds + B0OFF + (32 * mh + 8 * g + 4 * lh) * 4);
            bs0v[4 * g] = b4[0]; bs0v[4 * g + 1] = b4[1]; bs0v[4 * g + 2] = b4[2]; bs0v[4 * g + 3] = b4[3];
        }
        const LDS float *xs = reinterpret_cast<const LDS float *>(lds + XOFF);
#pragma unroll
        for (int rs = ph; rs < NRS; rs += 2) {
            const int r = 32 * rs + l31, p = tile * PBS - 3 + r;
            const bool inside = p >= 0 && p < L1; // (a row outside the read is layer 1's zero padding)
            cnn_h8 xh, xl;
            float mxx = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = 3 * p - 3 + e;
                float v = 0.0f;
                if (e < CNN_K) { v = xs[3 * r + e]; asm volatile("" : "+v"(v)); } // (read by every lane: a select, not a branch per sample)
                v = (e < CNN_K && lh == 0 && inside && i >= 0 && i < F.Lc) ? v : 0.0f;
                _Float16 hi, lo;
                cns_split(v, hi, lo);
                xh[e] = hi; xl[e] = lo;
                if (e < CNN_K) mxx = __builtin_fmaxf(mxx, __builtin_fabsf(v));
            }
            // a sample beyond the float16 range (or infinite): the call is repeated on the float32 kernels, like an activation beyond it.
            // (A NaN sample makes its rows zero here as there: max(NaN, 0) = 0.)
            if (mxx >= 32768.0f) hmax[0] = 0x7c00;
            // (a row outside the read: zero factors instead of a select per pair of values)
            const float k1 = inside ? k1_ : 0.0f, k2 = inside ? k2_ : 0.0f;
            cnn_f32x16 am = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xh, bs0v, 0, 0, 0);
            cnn_f32x16 ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, xh, zero16, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xl, ax, 0, 0, 0);
            LDS char *srow = lds + r * CNS_ROWB + (32 * mh + 4 * lh) * 2;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                cnn_h2 hq[2], lq[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++) {
                    const cnn_f2 a2 = {am[4 * g + 2 * q2], am[4 * g + 2 * q2 + 1]}, x2 = {ax[4 * g + 2 * q2], ax[4 * g + 2 * q2 + 1]};
                    cnn_f2 v = __builtin_elementwise_fma(x2, (cnn_f2){k2, k2}, a2 * k1);
                    v = __builtin_elementwise_max(v, (cnn_f2){0.f, 0.f});
                    const cnn_h2 hi = __builtin_convertvector(v, cnn_h2);
                    const cnn_f2 rs2 = (v - __builtin_convertvector(hi, cnn_f2)) * 2048.0f;
                    hq[q2] = hi; lq[q2] = __builtin_convertvector(rs2, cnn_h2);
                    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(cnn_us2, hi));
                }
                const cnn_h4 hh = {hq[0][0], hq[0][1], hq[1][0], hq[1][1]}, ll = {lq[0][0], lq[0][1], lq[1][0], lq[1][1]};
                *reinterpret_cast<LDS cnn_h4 *>(srow + 16 * g) = hh;
                *reinterpret_cast<LDS cnn_h4 *>(srow + 128 + 16 * g) = ll;
            }
        }
    };
    if (it < total) { if constexpr (FIRST) load_x(it); else dma(it, 0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ADP_PHASE_TIMING
    // (diagnostic build: cycles of wave 0 of workgroup 0 per phase of a step, summed over the launch: g_dbg[64] rows made / tile awaited,
    // [65] k-loop, [66] conversion, [67] layer 3 or copy-out, [68] steps; layers 2 + 3: 70 .. 74)
    auto cns_now = []() { long long t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
    long long ph_[6] = {0, 0, 0, 0, 0, 0}, pt_ = cns_now();
    long long nst_ = 0;
#define CNS_PH(k) do { const long long t_ = cns_now(); ph_[k] += t_ - pt_; pt_ = t_; } while (0)
#else
#define CNS_PH(k) do { } while (0)
#endif
    for (; it < total; it += gridDim.x) {
        // every wave has waited for its own share of tile `it` (before the loop / at the end of the previous step) and has
        // finished reading the other buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (FIRST) {
            // (one buffer: the staging rows of the step before have left it -- the barrier above -- and nothing else is in flight)
            make_rows(it);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + gridDim.x < total) load_x(it + gridDim.x); // the next tile's samples arrive during this tile's k-loop
            if constexpr (BS_LDS) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const cnn_f4 b4 = *reinterpret_cast<const LDS cnn_f4 *>(lds + B0OFF + (CNN_C + 32 * mh + 8 * g + 4 * lh) * 4);
                    bsv[4 * g] = b4[0]; bsv[4 * g + 1] = b4[1]; bsv[4 * g + 2] = b4[2]; bsv[4 * g + 3] = b4[3];
                }
            }
        }
        // the next tile's LDS-DMA pieces are issued INSIDE the k-loop, one per k-step (a piece costs ~60 cycles of issue among MFMAs,
        // 100-185 in a burst in front of them: MI355X_MICROARCH.md), still older than this step's stores for the counted wait below
        // (NT = 4 sits at the register limit -- 512 with the accumulators of four tiles: there the pieces stay a burst at the step's top)
        CNS_PH(0);
        constexpr bool INLOOP = NT < 4;
        if (!FIRST && !INLOOP && it + gridDim.x < total) dma(it + gridDim.x, buf ^ 1);
        const bool more = !FIRST && INLOOP && it + gridDim.x < total;
        const GLB char *nsrc = nullptr;
        if (more) {
            const int tix = it + gridDim.x, nn = tix / tiles_per_read, tile2 = tix - nn * tiles_per_read;
            nsrc = (const GLB char *)in + ((size_t)nn * Lrows + CNS_FRONT - 3 - (LAST ? 1 : 0) + (size_t)tile2 * PBS) * CNS_ROWB + lane * 16;
        }
        LDS char *ndst = lds + (buf ^ 1) * TILE_B;
        static_assert((NDMA + 3) / 4 <= CNS_KSTEPS, "one DMA piece per k-step and wave");
        // B fragment of k-step (t, cg), position tile j: row ph * 32 NT + 32 j + (lane & 31) + t, channels 16 cg + 8 (lane >> 5) ..
        const LDS char *tb = lds + buf * TILE_B + (ph * (NT * 32) + l31) * CNS_ROWB + lh * 16;
        // (no initialisation of the accumulators: the first k-step's MFMAs take the bias vector / zero as their C operand -- NT x 32
        // v_accvgpr_write per tile otherwise, in front of the MFMAs of the only wave on the SIMD)
        cnn_f32x16 am[NT], ax[NT];
        cnn_h8 fh[2][NT], fl[2][NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            fh[0][j] = *reinterpret_cast<const LDS cnn_h8 *>(tb + (32 * j) * CNS_ROWB);
            fl[0][j] = *reinterpret_cast<const LDS cnn_h8 *>(tb + (32 * j) * CNS_ROWB + 128);
        }
#pragma unroll
        for (int k = 0; k < CNS_KSTEPS; k++) {
            if (more && wave + 4 * k < NDMA)
                __builtin_amdgcn_global_load_lds((const GLB float *)(nsrc + (wave + 4 * k) * 1024), (LDS float *)(ndst + (wave + 4 * k) * 1024), 16, 0, 0);
            if (k + 1 < CNS_KSTEPS) {
                const int t1 = (k + 1) >> 2, cg1 = (k + 1) & 3;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    fh[(k + 1) & 1][j] = *reinterpret_cast<const LDS cnn_h8 *>(tb + (32 * j + t1) * CNS_ROWB + cg1 * 32);
                    fl[(k + 1) & 1][j] = *reinterpret_cast<const LDS cnn_h8 *>(tb + (32 * j + t1) * CNS_ROWB + cg1 * 32 + 128);
                }
            }
#if !(CNS_ABL & 1)
#pragma unroll
            for (int j = 0; j < NT; j++) am[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], fh[k & 1][j], k == 0 ? bsv : am[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; j++) ax[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], fh[k & 1][j], k == 0 ? zero16 : ax[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; j++) ax[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], fl[k & 1][j], ax[j], 0, 0, 0);
#else
#pragma unroll
            for (int j = 0; j < NT; j++) { if (k == 0) { am[j] = bsv; ax[j] = zero16; } am[j][k & 15] += (float)fh[k & 1][j][0] * (float)wh[k][0]; ax[j][k & 15] += (float)fl[k & 1][j][0] * (float)wl[k][0]; }
#endif
            asm volatile("" ::: "memory");
        }
        CNS_PH(1);
        const int n = it / tiles_per_read, tile = it - n * tiles_per_read;
        // EPILOGUE through LDS: the waves lay their split results down in the tile they have just consumed, in the rows' HBM
        // layout, and the workgroup copies the PB rows out as one contiguous range, 16 bytes per lane (a lane's own 8-byte
        // pieces reach the L2 as partial 16-byte writes from four instructions of two waves: 0.45 ms of a 2.5 ms layer).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // every wave has read its last fragment of this buffer
        asm volatile("" ::: "memory");
        LDS char *stg = lds + buf * TILE_B;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            LDS char *srow = stg + (ph * (NT * 32) + 32 * j + l31) * CNS_ROWB + (32 * mh + 4 * lh) * 2;
            // LAST: a row outside the read is layer 3's zero padding (its activation of the padded input is not zero)
            const int pos_ = tile * PBS - 1 + ph * (NT * 32) + 32 * j + l31;
            // (a select, not a factor: position -1 is computed from the dump row in front of the read; FIRST: the last six positions from rows nobody made)
            const bool keep_ = LAST ? (pos_ >= 0 && pos_ < L1) : FIRST ? (ph * (NT * 32) + 32 * j + l31 < PBS) : true;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // two values per instruction where the hardware has a packed form (v_pk_mul / v_pk_fma / v_cvt_pk_f16_f32 / v_pk_add):
                // one wave per SIMD -- nothing hides the epilogue's vector instructions
                cnn_h2 hq[2], lq[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const cnn_f2 a2 = {am[j][4 * g + 2 * q], am[j][4 * g + 2 * q + 1]}, x2 = {ax[j][4 * g + 2 * q], ax[j][4 * g + 2 * q + 1]};
                    cnn_f2 v = __builtin_elementwise_fma(x2, (cnn_f2){cx, cx}, a2 * inv_sw);
                    v = __builtin_elementwise_max(v, (cnn_f2){0.f, 0.f}); // (a NaN becomes 0, as `v > 0 ? v : 0` makes it in k_cnn_conv64)
                    if ((LAST || FIRST) && !keep_) v = (cnn_f2){0.f, 0.f};
                    const cnn_h2 hi = __builtin_convertvector(v, cnn_h2);
                    const cnn_f2 rs = (v - __builtin_convertvector(hi, cnn_f2)) * 2048.0f;
                    hq[q] = hi; lq[q] = __builtin_convertvector(rs, cnn_h2);
                    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(cnn_us2, hi));
                }
                const cnn_h4 hh = {hq[0][0], hq[0][1], hq[1][0], hq[1][1]}, ll = {lq[0][0], lq[0][1], lq[1][0], lq[1][1]};
                *reinterpret_cast<LDS cnn_h4 *>(srow + 16 * g) = hh;
                *reinterpret_cast<LDS cnn_h4 *>(srow + 128 + 16 * g) = ll;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        CNS_PH(2);
        if constexpr (LAST) {
            // ---- layer 3 on the tile's rows in LDS: wave q takes the 32-position subtiles q, q + 4, ... of the tile's 2 NT
            LDS float *psum = (LDS float *)(lds + 2 * TILE_B); // [o 2][t 7][position PB] partial sums of the taps
            if constexpr (!W3_RESIDENT) {
                const cnn_h8 *wp = reinterpret_cast<const cnn_h8 *>(w3sp) + lane;
#pragma unroll
                for (int cg = 0; cg < 4; cg++) { ah[cg] = wp[(size_t)(cg * 2) * 64]; al[cg] = wp[(size_t)(cg * 2 + 1) * 64]; }
            }
            for (int st = wave; st < 2 * NT; st += 4) {
                const LDS char *rb = stg + (32 * st + l31) * CNS_ROWB + lh * 16;
                cnn_f32x16 pm = zero16, px = zero16;
#pragma unroll
                for (int cg = 0; cg < 4; cg++) {
                    const cnn_h8 bh = *reinterpret_cast<const LDS cnn_h8 *>(rb + cg * 32), bl = *reinterpret_cast<const LDS cnn_h8 *>(rb + cg * 32 + 128);
                    pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cg], bh, pm, 0, 0, 0);
                    px = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cg], bh, px, 0, 0, 0);
                    px = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cg], bl, px, 0, 0, 0);
                }
                // lane (position 32 st + l31, output channel lh): registers 0 .. 6 are the taps' partial sums (k_cns_split_w3)
#pragma unroll
                for (int t = 0; t < CNN_K; t++) psum[(lh * CNN_K + t) * PB + 32 * st + l31] = __builtin_fmaf(px[t], 1.0f / 2048.0f, pm[t]) * inv_s3;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CNS_PH(4); // (layer 3's GEMM and its partial sums; the rest of this branch -- the combination and the score stores -- stays in [3])
            // outputs 3 jg + d of the tile's interior positions jg = tile * PBS + (j - 1), j = 1 .. PB - 2.  Every thread issues the same
            // number of stores (those without an output go to the dump row of `out`, which this layer does not write otherwise), so that
            // the counted wait below leaves them in flight and only makes sure of the next tile's DMA, which is older
            constexpr int NIT = (2 * PBS + 255) / 256;
            float *dumpf = reinterpret_cast<float *>(out) + lane;
#pragma unroll
            for (int k2 = 0; k2 < NIT; k2++) {
                const int i = k2 * 256 + (int)threadIdx.x;
                const bool have = i < 2 * PBS;
                const int o = have ? i / PBS : 0, j = have ? i - o * PBS + 1 : 1;
                const int jg = tile * PBS + j - 1;
                const bool ok = have && jg < L1;
                const LDS float *pq = psum + (o * CNN_K) * PB + j;
                const float bo = o ? b3_1 : b3_0; // (fetched once: a load here is a memory round trip in every tile's last phase)
                const float y0 = bo + ((pq[6 * PB - 1] + pq[3 * PB]) + pq[0 * PB + 1]);
                const float y1 = bo + (pq[4 * PB] + pq[1 * PB + 1]);
                const float y2 = bo + (pq[5 * PB] + pq[2 * PB + 1]);
                float *so = scores + ((size_t)n * 2 + o) * Lo + 3 * (size_t)(ok ? jg : 0);
                *((ok && 3 * jg < Lo) ? so : dumpf) = y0;
                *((ok && 3 * jg + 1 < Lo) ? so + 1 : dumpf) = y1;
                *((ok && 3 * jg + 2 < Lo) ? so + 2 : dumpf) = y2;
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NIT) : "memory");
        } else {
        {
            // rows tile * PB .. of read n; pieces of rows at or beyond L1 go to the read's row 0 (which nothing reads), so that
            // every store is issued and the count below stays exact
            const int nvalid = (L1 - tile * PBS < PBS ? L1 - tile * PBS : PBS) * 17;
            char *rbase = reinterpret_cast<char *>(out) + (size_t)n * Lrows * CNS_ROWB;
            char *obase = rbase + (size_t)(CNS_FRONT + tile * PBS) * CNS_ROWB;
            cnn_h8 piece[NSTORE];
#pragma unroll
            for (int c = 0; c < NSTORE; c++) {
                const int i = c * 256 + (int)threadIdx.x;
                piece[c] = *reinterpret_cast<const LDS cnn_h8 *>(stg + (i < PB * 17 ? i : 0) * 16);
            }
#pragma unroll
            for (int c = 0; c < NSTORE; c++) {
                const int i = c * 256 + (int)threadIdx.x;
                char *dst = i < nvalid ? obase + (size_t)i * 16 : rbase + (i % 17) * 16;
#if (CNS_ABL & 2)
                if (piece[c][0] == (_Float16)12345.f && piece[c][1] == (_Float16)77.f)
#endif
                *reinterpret_cast<cnn_h8 *>(dst) = piece[c];
            }
        }
        // the DMA issued at the top of this step is older than these NSTORE stores
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
        }
        if (!FIRST) buf ^= 1;
        CNS_PH(3);
#ifdef ADP_PHASE_TIMING
        nst_++;
#endif
    }
#ifdef ADP_PHASE_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        constexpr int slot[2][5] = {{64, 65, 66, 67, 68}, {70, 71, 72, 73, 74}}; // (layer 1 / layer 2)
        for (int k = 0; k < 4; k++) atomicAdd(&g_dbg[slot[LAST][k]], (unsigned long long)ph_[k]);
        if (LAST) atomicAdd(&g_dbg[75], (unsigned long long)ph_[4]);
        atomicAdd(&g_dbg[slot[LAST][4]], (unsigned long long)nst_);
    }
#endif
    const bool bad = hmax[0] >= 0x7800 || hmax[1] >= 0x7800;
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// (Round 4, measured and dropped: the same layers with two waves per SIMD -- k_cnn_conv64s8, no gain -- and with a tile's epilogue under
// the next tile's MFMAs -- k_cnn_conv64p, slower; layer 0 computed inside layer 1's kernel as float32 fmaf chains -- no gain (round 5's
// FIRST does it on the matrix cores).  Sources and records: tools/experiments/r05_pruned_variants.patch,
// profiles/r04_tried_and_dropped.txt.)

// ---------------------------------------------------------------- layer 3: ConvTranspose1d(64 -> 2, k 7, stride 3, pad 3) from split rows
// the arithmetic of k_cnn_conv_out on a = hi + lo 2^-11; thread j makes the outputs 3 j, 3 j + 1, 3 j + 2 of both channels.
// grid = (ceil(L1 / 128), n); block = 128: the 130 rows of the block are one contiguous range, brought into LDS by LDS-DMA
// (coalesced) and read from there row by row (272-byte pitch: conflict-free 16-byte reads).
#define CNS_OUT_P 128
#define CNS_OUT_LDS (((CNS_OUT_P + 2) * CNS_ROWB + 1023) / 1024 * 1024)
__global__ void __launch_bounds__(CNS_OUT_P) k_cnn_conv_out_s(const _Float16 *__restrict__ h, int L1, int Lrows, int Lo,
                                                               const float *__restrict__ w /* [64][2][7] times 1 / CNS_ASCALE */, const float *__restrict__ b,
                                                               float *__restrict__ scores /* [n][2][Lo] */)
{
    __shared__ __attribute__((aligned(16))) char rows_[CNS_OUT_LDS];
    LDS char *rows = (LDS char *)rows_;
    const int n = blockIdx.y;
    const int j0 = blockIdx.x * CNS_OUT_P;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const GLB char *src = (const GLB char *)h + ((size_t)n * Lrows + CNS_FRONT + j0 - 1) * CNS_ROWB + lane * 16;
        for (int inst = wave; inst < CNS_OUT_LDS / 1024; inst += CNS_OUT_P / 64)
            __builtin_amdgcn_global_load_lds((const GLB float *)(src + inst * 1024), (LDS float *)(rows + inst * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= L1) return;
    const LDS char *hr = rows + (threadIdx.x + 1) * CNS_ROWB;
    float y[2][3];
#pragma unroll
    for (int o = 0; o < 2; o++) { y[o][0] = b[o]; y[o][1] = b[o]; y[o][2] = b[o]; }
#pragma unroll 2
    for (int c8 = 0; c8 < CNN_C / 8; c8++) {
        const cnn_h8 mh_ = *reinterpret_cast<const LDS cnn_h8 *>(hr - CNS_ROWB + c8 * 16), ml_ = *reinterpret_cast<const LDS cnn_h8 *>(hr - CNS_ROWB + 128 + c8 * 16);
        const cnn_h8 zh_ = *reinterpret_cast<const LDS cnn_h8 *>(hr + c8 * 16), zl_ = *reinterpret_cast<const LDS cnn_h8 *>(hr + 128 + c8 * 16);
        const cnn_h8 ph_ = *reinterpret_cast<const LDS cnn_h8 *>(hr + CNS_ROWB + c8 * 16), pl_ = *reinterpret_cast<const LDS cnn_h8 *>(hr + CNS_ROWB + 128 + c8 * 16);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            // (the activations carry CNS_ASCALE; its inverse sits in the weights -- w holds w3 / CNS_ASCALE: a power of two moved from one
            // factor of every product to the other, the same real products and so the same fmaf results, three multiplies per channel fewer)
            const float hm = cns_join(mh_[e], ml_[e]), h0 = cns_join(zh_[e], zl_[e]), hp = cns_join(ph_[e], pl_[e]);
            const float *wc = w + (c8 * 8 + e) * 2 * CNN_K;
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const float *wo = wc + o * CNN_K;
                y[o][0] = __builtin_fmaf(hm, wo[6], y[o][0]);
                y[o][0] = __builtin_fmaf(h0, wo[3], y[o][0]);
                y[o][0] = __builtin_fmaf(hp, wo[0], y[o][0]);
                y[o][1] = __builtin_fmaf(h0, wo[4], y[o][1]);
                y[o][1] = __builtin_fmaf(hp, wo[1], y[o][1]);
                y[o][2] = __builtin_fmaf(h0, wo[5], y[o][2]);
                y[o][2] = __builtin_fmaf(hp, wo[2], y[o][2]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
        float *so = scores + ((size_t)n * 2 + o) * Lo + 3 * j;
#pragma unroll
        for (int d = 0; d < 3; d++) if (3 * j + d < Lo) so[d] = y[o][d];
    }
}
