// cnn_conv.h -- C2: the CNN boundary head, hand-written for gfx950.
//
// reference adapted/detect/cnn.py:16-52 (BoundariesCNN) and :85-98 (cnn_score):
//   Conv1d(1 -> 64, k 7, stride 3, pad 3) - ReLU - Conv1d(64 -> 64, k 7, pad 3) - ReLU - Conv1d(64 -> 64, k 7, pad 3) - ReLU -
//   ConvTranspose1d(64 -> 2, k 7, stride 3, pad 3);   float32 throughout.
// Lengths: Lc -> L1 = (Lc - 1) / 3 + 1 -> L1 -> L1 -> Lo = 3 L1 - 2   (20 050 -> 6 684 -> 20 050 at the 200 k window).
//
// 98 % of the arithmetic (2 x 64 x 64 x 7 MAC per position) is in the two 64 -> 64 layers: a dense contraction
// out[o, p] = sum_{c, t} W[o, c, t] * in[c, p + t - 3], i.e. a GEMM with M = 64 output channels, N = positions, K = 448,
// run on the float32 matrix cores (v_mfma_f32_32x32x2_f32: exact float32, an fmaf chain in k order -- the accumulation
// order is fixed by this file, so scores are reproducible run to run).  Layout of the work:
//
//   * WEIGHTS STAY IN REGISTERS.  A wave owns 32 output channels; its A fragments for all 224 k-steps (7 taps x 32 channel
//     pairs) are 224 VGPRs loaded once per kernel.  One wave per SIMD (the register file is the budget, not occupancy).
//   * ACTIVATIONS GO THROUGH LDS.  A workgroup (4 waves = 2 channel halves x 2 position halves) computes 64 channels x
//     PB = 64 NT positions per step from an LDS image [64 channels][PB + 8] of the input, filled by LDS-DMA
//     (global_load_lds_dwordx4, no staging registers) one step ahead into the second buffer.  A B fragment is one
//     ds_read_b32: lanes 0-31 read 32 consecutive positions of channel c, lanes 32-63 of channel c + 1 (the two 32-lane
//     groups never conflict, consecutive addresses within a group hit 32 different banks).
//   * activation buffers in HBM are [read][64][Lpad] with CNN_PADL zero columns in front of position 0 and at least 4
//     behind the last tile, so that the "same" padding of the convolutions is read, not branched on, and every tile
//     row starts 16-byte aligned.  Kernels only ever write positions < L1: the zeros are laid down once per buffer.
//
// The first layer (448 MAC per position) and the transposed last layer (~900 MAC per input position) are plain VALU
// kernels bound by the 1.7 MB per read they write / read.
#pragma once
#include "common.h"

#define CNN_C 64
#define CNN_K 7
#define CNN_PADL 4

typedef float cnn_f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- layer 0: Conv1d(1 -> 64, k 7, stride 3, pad 3) + ReLU
// grid = (ceil(L1 / 256), n); block = 256: one output position per thread, the 64 channels in turn (weights are
// wave-uniform: scalar loads); stores of a channel are coalesced over the positions.
__global__ void __launch_bounds__(256) k_cnn_conv_in(const float *__restrict__ x, int Lc, int L1, int Lpad,
                                                      const float *__restrict__ w /* [64][1][7] */, const float *__restrict__ b,
                                                      float *__restrict__ out)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= L1) return;
    const float *row = x + (size_t)n * Lc;
    float v[CNN_K];
#pragma unroll
    for (int t = 0; t < CNN_K; t++) { const int i = 3 * p + t - 3; v[t] = (i >= 0 && i < Lc) ? row[i] : 0.f; }
    float *o = out + (size_t)n * CNN_C * Lpad + CNN_PADL + p;
#pragma unroll 4
    for (int c = 0; c < CNN_C; c++) {
        float acc = b[c];
#pragma unroll
        for (int t = 0; t < CNN_K; t++) acc = __builtin_fmaf(w[c * CNN_K + t], v[t], acc);
        o[(size_t)c * Lpad] = acc > 0.f ? acc : 0.f;
    }
}

// ---------------------------------------------------------------- layers 1, 2: Conv1d(64 -> 64, k 7, pad 3) + ReLU on the matrix cores
// grid = persistent (one workgroup per CU); block = 256; dynamic LDS = 2 * 64 * (64 NT + 8) floats.
template <int NT>
__global__ void __launch_bounds__(256, 1) k_cnn_conv64(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ w /* [64][64][7] */, const float *__restrict__ bias,
                                                       int n_reads, int L1, int Lpad, int tiles_per_read)
{
    constexpr int PB = 64 * NT, S = PB + 8, TILE = CNN_C * S, NDMA = TILE / 256;
    extern __shared__ __attribute__((aligned(16))) float cnn_lds_raw[];
    LDS float *lds = (LDS float *)cnn_lds_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mh = wave & 1, ph = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // A fragments of every k-step: lane l holds W[o = 32 mh + (l & 31)][c = 2 cp + (l >> 5)][t]
    float wr[CNN_K][32];
    {
        const float *wl = w + ((size_t)(32 * mh + l31) * CNN_C + lh) * CNN_K;
#pragma unroll
        for (int t = 0; t < CNN_K; t++)
#pragma unroll
            for (int cp = 0; cp < 32; cp++) wr[t][cp] = wl[(2 * cp) * CNN_K + t];
    }
    // C/D layout of the 32x32 tile: column = lane & 31 (position), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bs[r] = bias[32 * mh + (r & 3) + 8 * (r >> 2) + 4 * lh];

    const long long total = (long long)n_reads * tiles_per_read;
    auto dma = [&](long long tix, int b) {
        const int n = (int)(tix / tiles_per_read), tile = (int)(tix - (long long)n * tiles_per_read);
        const GLB float *src0 = (const GLB float *)in + (size_t)n * CNN_C * Lpad + (size_t)tile * PB;
        for (int inst = wave; inst < NDMA; inst += 4) {
            const int e = inst * 256 + lane * 4;
            const int c = e / S, col = e - c * S;
            __builtin_amdgcn_global_load_lds(src0 + (size_t)c * Lpad + col, lds + b * TILE + inst * 256, 16, 0, 0);
        }
    };
    long long it = blockIdx.x;
    int buf = 0;
    if (it < total) dma(it, 0);
    for (; it < total; it += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of tile `it` has landed (and its stores are out)
        __syncthreads();                                   // ... everybody's has; nobody reads the other buffer any more
        if (it + gridDim.x < total) dma(it + gridDim.x, buf ^ 1);
        const LDS float *tb = lds + buf * TILE + lh * S + ph * (NT * 32) + l31 + 1;
        cnn_f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][r] = bs[r];
        // k-step = (tap t, channel pair cp): NT MFMAs on NT B fragments.  The fragments of step k + 1 are requested before
        // the MFMAs of step k (an LDS read takes ~64-128 cycles, a step NT x 64); the empty asm keeps the compiler from
        // hoisting hundreds of reads ahead (it spills the weights otherwise).
        float bq[2][NT];
#pragma unroll
        for (int j = 0; j < NT; j++) bq[0][j] = tb[32 * j];
#pragma unroll
        for (int k = 0; k < CNN_K * 32; k++) {
            if (k + 1 < CNN_K * 32) {
                const int t1 = (k + 1) >> 5, cp1 = (k + 1) & 31;
#pragma unroll
                for (int j = 0; j < NT; j++) bq[(k + 1) & 1][j] = tb[(2 * cp1) * S + 32 * j + t1];
            }
#pragma unroll
            for (int j = 0; j < NT; j++)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[k >> 5][k & 31], bq[k & 1][j], acc[j], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
        const int n = (int)(it / tiles_per_read), tile = (int)(it - (long long)n * tiles_per_read);
        float *orow = out + ((size_t)n * CNN_C + 32 * mh + 4 * lh) * Lpad + CNN_PADL;
        const int p0 = tile * PB + ph * (NT * 32) + l31;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int p = p0 + 32 * j;
            if (p < L1) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = acc[j][r];
                    orow[(size_t)((r & 3) + 8 * (r >> 2)) * Lpad + p] = v > 0.f ? v : 0.f;
                }
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------- layer 3: ConvTranspose1d(64 -> 2, k 7, stride 3, pad 3)
// y[o][q] = b[o] + sum_c sum_p h[c][p] w[c][o][q - 3 p + 3]  (0 <= q - 3 p + 3 <= 6).  Thread j makes the outputs q = 3 j,
// 3 j + 1, 3 j + 2 of both channels from h[:, j - 1], h[:, j], h[:, j + 1] (zeros outside [0, L1) come from the padding):
//   q = 3 j     : taps 6, 3, 0 on p = j - 1, j, j + 1;   q = 3 j + 1 : taps 4, 1 on p = j, j + 1;   q = 3 j + 2 : taps 5, 2.
// grid = (ceil(L1 / 256), n); block = 256.
__global__ void __launch_bounds__(256) k_cnn_conv_out(const float *__restrict__ h, int L1, int Lpad, int Lo,
                                                       const float *__restrict__ w /* [64][2][7] */, const float *__restrict__ b,
                                                       float *__restrict__ scores /* [n][2][Lo] */)
{
    const int n = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= L1) return;
    const float *hr = h + (size_t)n * CNN_C * Lpad + CNN_PADL + j;
    float y[2][3];
#pragma unroll
    for (int o = 0; o < 2; o++) { y[o][0] = b[o]; y[o][1] = b[o]; y[o][2] = b[o]; }
#pragma unroll 4
    for (int c = 0; c < CNN_C; c++) {
        const float hm = hr[(size_t)c * Lpad - 1], h0 = hr[(size_t)c * Lpad], hp = hr[(size_t)c * Lpad + 1];
        const float *wc = w + c * 2 * CNN_K;
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
#pragma unroll
    for (int o = 0; o < 2; o++) {
        float *so = scores + ((size_t)n * 2 + o) * Lo + 3 * j;
#pragma unroll
        for (int d = 0; d < 3; d++) if (3 * j + d < Lo) so[d] = y[o][d];
    }
}
