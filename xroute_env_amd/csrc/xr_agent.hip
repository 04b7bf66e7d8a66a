// xr_agent.hip — the obstacle tower of the reference's RepresentationNetwork as ONE fused gfx950 kernel (SURVEY.md §8 row f1,
// a CONSUMER of the env path: with the DQN / PPO counterpart attached, the tower's convolutions through MIOpen cost 15x the env step).
//
// What it computes, per env (reference baseline/baseline_utils.py:231-379, `ob_conv1 -> ob_align_conv1 -> ob_conv2 -> ob_align_conv2`,
// eval mode, BatchNorm folded into the convolutions by the caller — xroute_env_amd/agents.py FusedObstacleTower):
//   x  [D,H,W]            plane 0 of the env's observation
//   a  = relu(conv3(relu(conv3(x))) + x)                                   ResidualBlock(1), 3x3x3, zero padding 1
//   b  = conv5(a), 1 -> 7 channels, stride (sd,sh,sw), padding 1           [7,od,oh,ow]   (no activation)
//   P  = b zero-padded at the far ends to the standard grid [7,3,64,64]
//   c  = relu(conv3(relu(conv3(P))) + P)                                   ResidualBlock(7)
//   v[w] = bias + sum_{ch,d,h,kw} Wal2[ch,d,h,kw] * c[ch,d,h,w+kw-1]       kernel (3,64,3), padding (0,0,1): a 64-vector
// Outside the cells the data can influence (h < oh + 2, w < ow + 2) c equals the block's response to an all-zero grid, which does not
// depend on the env: the caller folds that part (and the bias) into `kvec`, the kernel sums the inside cells only.
//
// One workgroup of 1024 threads per env, everything in LDS (b: 7*od*oh*ow floats, the first activation of the 7-channel block:
// 7*3*(oh+3)*(ow+3) floats, both channel-interleaved; the 1-channel stages live inside that space on zero-padded grids of an odd row pitch).
// Stages: the 1-channel block on strips of cells (lanes on consecutive rows: no bank conflicts), its output in place of x; the aligning
// convolution on packed FMAs with its weights through scalar loads (every lane of a wave uses the same weight); the two 7 -> 7-channel
// convolutions as a paired implicit GEMM on v_mfma_f32_16x16x4_f32 (two output columns per instruction, weights as per-lane A operands
// in VGPRs); the last convolution accumulated straight from the accumulators.  fp32 throughout; every sum runs in a fixed order (no
// atomics): same input, same bits.  Stage cycles: `make ttiming` + tools/tower_probe.py (XT_PHASES=1); DESIGN.md §7.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/xroute_hip.h"
#include "xr_device.h"

namespace {

typedef float xt_f2 __attribute__((ext_vector_type(2)));     // pairs of output channels: v_pk_fma_f32
typedef float xt_f4 __attribute__((ext_vector_type(4)));     // accumulator of v_mfma_f32_16x16x4_f32

struct XtDims {
    int D, H, W;          // input grid
    int sd, sh, sw;       // stride of the aligning convolution
    int od, oh, ow;       // its output
    int cols;             // columns of the last stage: ow + 2
    int y_in_b;           // the 1-channel block's intermediate lives in b's LDS space (else behind x in the first activation's space, which then grows)
    int strip;            // cells per thread of the 1-channel convolutions: 3, 5 or 4 (the packed form), whichever takes fewer instructions over all passes
    int vec_load;         // rows of x are multiples of 16 bytes at 16-byte aligned addresses: four cells per load
    int tail;             // floats of b + the first activation's allocation: behind them 8 zero words and the waves' column sums [nw][cols + 2][3]
};

// packed weights (floats), offsets
constexpr int XT_A1 = 0;                    // 27 + 1     block(1).conv1 (BatchNorm folded)
constexpr int XT_A2 = XT_A1 + 28;           // 27 + 1     block(1).conv2
constexpr int XT_AL1 = XT_A2 + 28;          // 125*8 + 8  align1: [kd][kh][kw][co padded to 8], bias[8]
constexpr int XT_C1 = XT_AL1 + 1008;        // 16*64*4 + 8 block(7).conv1 as the A operands of its 63 matrix instructions: [step / 4][lane][step % 4] (see xt_load_wA), bias[8]
constexpr int XT_C2 = XT_C1 + 4104;         // same       block(7).conv2
constexpr int XT_AL2 = XT_C2 + 4104;        // 3*64*8*3   align2: [d][h][ch padded to 8][kw]: the 12 weights a lane of the last stage needs are neighbours
constexpr int XT_KV = XT_AL2 + 4608;        // 64         bias + contribution of every cell the data cannot influence
constexpr int XT_C1B = XT_KV + 64;          // 9*2*64*4   block(7).conv1 as bf16 A fragments of v_mfma_f32_16x16x32_bf16 (matrix mode 1, see xt_mm3b): [(kd, kh)][hi / lo][lane][4 words]
constexpr int XT_C2B = XT_C1B + 4608;       // same       block(7).conv2
constexpr int XT_AL1B = XT_C2B + 4608;      // 5*2*2*64*4  align1 as split-bf16 fragments (matrix mode 1, strides (sd, 1, 1)): [kd][g][kind][lane][4 words]
constexpr int XT_TOTAL = XT_AL1B + 5120;
// the NET tower (same 7-channel block + last convolution, its own weights in the XT_C1 .. XT_KV slots; XT_A1 .. XT_AL1 unused) has a sparse front end
// behind them (agents.FusedNetTower.pack):
constexpr int XN_A0 = XT_TOTAL;             // 27*8      block(7).conv1, input plane 0 (the access-point mask): [tap][co padded to 8]
constexpr int XN_AS = XN_A0 + 216;          // 27*8      the same, planes 1..6 summed (they are ONE aliased plane, baseline/build_3Dgrid.py:125)
constexpr int XN_BA = XN_AS + 216;          // 8         its bias (BatchNorm folded)
constexpr int XN_WB = XN_BA + 8;            // 27*7*8    block(7).conv2: [tap][ci][co padded to 8]
constexpr int XN_ZBG = XN_WB + 1512;        // 64*8      conv2 of the background relu(bias) + its bias, by which taps fall outside the grid:
                                            //           class = md << 4 | mh << 2 | mw, m = (coordinate > 0) | (coordinate < dim - 1) << 1
constexpr int XN_WL = XN_ZBG + 512;         // 125*7*8   align1: [tap][ci][co padded to 8]
constexpr int XN_WSUM = XN_WL + 7000;       // 2 (+6)    sum over align1's (then: block conv2's) taps and input channels of max_co |W|: bound every partial sum of the two scatters per unit of |o1 - bg1| (|dy|)
constexpr int XN_TOTAL = XN_WSUM + 8;

// what the net tower's front end reads besides the weights
struct XnArgs {
    const XrRegionDev* regions;
    const int32_t* net_csr;
    const int32_t* ap_feat;         // node index | "has a same-net axis neighbour" << 31, per access point (xr_batch.cpp)
    const int32_t* pair_region;
    const int32_t* pair_net;        // 1-based
    const float* bg;                // align1(block(0)) of this grid shape: [od][oh][ow][7]
    int32_t* flags;                 // per pair: 0 ok, 1 lists do not fit (caller: framework path), 2 another shape / no such net
    int32_t n_regions;
};

// A operand of the paired implicit GEMM (see the 7-channel block below): lane l holds row m = l & 15 = (dw, co) and k = l >> 4 = c4 of every
// one of the 63 (kd, kh, ci) steps: W[co][ci][kd][kh][kw = c4 - dw], 0 where kw is no tap or co is the padding channel — laid out by the caller
// (agents.FusedObstacleTower) as [step / 4][lane][step % 4]: 16 coalesced 16-byte loads per lane (as 63 gathers from the convolution's own
// layout the fetch took 15 k cycles per convolution and workgroup: a tenth of the kernel).
__device__ __forceinline__ void xt_load_wA(const float* __restrict__ wc, int lane, float (&wA)[64]) {
    const xt_f4* __restrict__ src = reinterpret_cast<const xt_f4*>(wc) + lane;
#pragma unroll
    for (int s4 = 0; s4 < 16; s4++) {
        const xt_f4 v = src[s4 * 64];
        wA[s4 * 4 + 0] = v[0]; wA[s4 * 4 + 1] = v[1]; wA[s4 * 4 + 2] = v[2]; wA[s4 * 4 + 3] = v[3];
    }
}

// ---- matrix mode 1 (round 6): the 7 -> 7-channel convolutions on v_mfma_f32_16x16x32_bf16 with SPLIT operands — every fp32 value v is held as two bf16,
// hi = bf16(v) and lo = bf16(v - hi) (v - hi - lo <= 2^-17 |v|), and a product is three instructions, hi x hi + hi x lo + lo x hi, accumulated in fp32 (the dropped
// lo x lo term is <= 2^-16 of the product).  One instruction takes K = 32 = the four input columns x eight channel slots of one (kd, kh) tap row — what the fp32
// form (K = 4) needs seven instructions for, at 32 cycles each against ~17 — so a tap row costs 3 x 17 instead of 7 x 32 cycles of the matrix pipe.  Lane l holds, for
// BOTH operands, the k-values of its quarter q = l >> 4 in the same order, so the sum over k does not care how the hardware numbers them; C / D are laid out like
// the fp32 form's.  Activations live in LDS as 16 + 16 bits per value (same footprint as fp32), a 7-channel cell in the order its readers want (xt_bfrag below): a lane
// reads its cell's seven words as before and they ARE the hi and the lo fragment, up to a shift and a mask of the seventh.  Measured against the fp32 form on the reference fixtures (CPU
// emulation first, tests/test_agents.py): normalised vectors within 1.1e-5 — the noise of a different fp32 summation order is 2e-6 … 5e-6.
typedef __bf16 xt_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int xt_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xt_bf16_rne(float x) {           // the bf16 nearest to x (ties to even), in the UPPER half of the word
    return (uint32_t)__builtin_bit_cast(unsigned short, static_cast<__bf16>(x)) << 16;      // (v_cvt_pk_bf16_f32: the integer form took four instructions)
}
__device__ __forceinline__ float xt_pack(float x) {                  // hi << 16 | lo as the bits of a float: a value of the 1-channel grid (align1's raw B operand)
    const uint32_t hi = xt_bf16_rne(x);
    const uint32_t lo = xt_bf16_rne(x - __uint_as_float(hi)) >> 16;
    return __uint_as_float(hi | lo);
}

// A fragments of one convolution: 9 tap rows x (hi, lo) x 4 words per lane, packed by the caller (agents._mfma_operand_bf16)
__device__ __forceinline__ void xt_load_wAb(const float* __restrict__ wc, int lane, xt_u4 (&ah)[9], xt_u4 (&al)[9]) {
    const xt_u4* __restrict__ src = reinterpret_cast<const xt_u4*>(wc) + lane;
#pragma unroll
    for (int s = 0; s < 9; s++) { ah[s] = src[(2 * s) * 64]; al[s] = src[(2 * s + 1) * 64]; }
}

// A 7-channel CELL of the matrix stages' LDS arrays (matrix mode 1), seven words, laid out for the reader: words 0..2 = the hi halves of the channel pairs
// (0,1) (2,3) (4,5) (even channel in the low 16 bits), words 3..5 = their lo halves, word 6 = channel 6 as hi << 16 | lo.  The hi and lo B fragments of a cell
// are then its words {0, 1, 2, w6 >> 16} and {3, 4, 5, w6 & 0xFFFF}: seven reads and TWO vector instructions (one word hi << 16 | lo per channel took six
// byte permutes on top: a cell is written once and read as a fragment ~30 times, and the 7 -> 7 stages are bound by the building of their fragments).
__device__ __forceinline__ void xt_bfrag(const float* __restrict__ cell, xt_u4& bh, xt_u4& bl) {
    const uint32_t* __restrict__ pw = reinterpret_cast<const uint32_t*>(cell);
    const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2], w3 = pw[3], w4 = pw[4], w5 = pw[5], w6 = pw[6];
    bh = xt_u4{w0, w1, w2, w6 >> 16};
    bl = xt_u4{w3, w4, w5, w6 & 0xFFFFu};
}
// the writer's side: `half` 0 = channels 0..3 (v0..v3) -> words 0, 1, 3, 4;  half 1 = channels 4..6 (v0..v2) -> words 2, 5, 6.  (A lane of the matrix stages'
// epilogues holds one half of a cell.)
__device__ __forceinline__ void xt_store_half(float* __restrict__ cell, const int half, const float v0, const float v1, const float v2, const float v3) {
    uint32_t* __restrict__ pw = reinterpret_cast<uint32_t*>(cell);
    const uint32_t h0 = xt_bf16_rne(v0), h1 = xt_bf16_rne(v1), h2 = xt_bf16_rne(v2), h3 = xt_bf16_rne(v3);
    const uint32_t l0 = xt_bf16_rne(v0 - __uint_as_float(h0)), l1 = xt_bf16_rne(v1 - __uint_as_float(h1));
    const uint32_t l2 = xt_bf16_rne(v2 - __uint_as_float(h2)), l3 = xt_bf16_rne(v3 - __uint_as_float(h3));
    pw[2 * half] = (h0 >> 16) | h1;
    pw[3 + 2 * half] = (l0 >> 16) | l1;
    pw[half ? 6 : 1] = half ? (h2 | (l2 >> 16)) : ((h2 >> 16) | h3);
    if (!half) pw[4] = (l2 >> 16) | l3;
}
__device__ __forceinline__ void xt_store_cell(float* __restrict__ cell, const float (&v)[7]) {
    xt_store_half(cell, 0, v[0], v[1], v[2], v[3]);
    xt_store_half(cell, 1, v[4], v[5], v[6], 0.f);
}
// ... and the values back (the residual of the block's second convolution): the half's channels as floats, hi + lo
__device__ __forceinline__ void xt_load_half(const float* __restrict__ cell, const int half, float (&v)[4]) {
    const uint32_t* __restrict__ pw = reinterpret_cast<const uint32_t*>(cell);
    const uint32_t a = pw[2 * half], c = pw[3 + 2 * half], b = pw[half ? 6 : 1], d = pw[half ? 6 : 4];
    v[0] = __uint_as_float(a << 16) + __uint_as_float(c << 16);
    v[1] = __uint_as_float(a & 0xFFFF0000u) + __uint_as_float(c & 0xFFFF0000u);
    v[2] = half ? __uint_as_float(b & 0xFFFF0000u) + __uint_as_float(b << 16) : __uint_as_float(b << 16) + __uint_as_float(d << 16);
    v[3] = half ? 0.f : __uint_as_float(b & 0xFFFF0000u) + __uint_as_float(d & 0xFFFF0000u);
}
// the input-slice-major form of the 7 -> 7 stages (a cell row's fragments built once for the up to three output slices it is a tap of) holds three accumulators
// beside the 72 weight registers: stages 24 k -> 18 k cycles.  The first stage takes it in both variants; the second only in the net variant — in the obstacle
// variant its epilogue's registers on top spill all 72 weights (stage 79 k), and with 768 threads (no spill) the epilogue's align2 weights can no longer be asked
// for ahead of a slice's matrix instructions (31 k)
template <int N> struct xt_ic { static constexpr int value = N; };
#ifndef XT_AL1_TC
#define XT_AL1_TC 4
#endif
template <bool NET, int MM> constexpr bool xt_reuse_b() { return NET && MM != 0; }
template <bool NET, int MM, int BT> constexpr bool xt_reuse_b2() { return MM != 0 && (NET || BT <= 768); }      // the second stage: see the launcher's thread count
__device__ __forceinline__ xt_f4 xt_mm3b(const xt_u4 ah, const xt_u4 al, const xt_u4 bh, const xt_u4 bl, xt_f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xt_bf8, ah), __builtin_bit_cast(xt_bf8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xt_bf8, ah), __builtin_bit_cast(xt_bf8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xt_bf8, al), __builtin_bit_cast(xt_bf8, bh), acc, 0, 0, 0);
    return acc;
}
// one tap row: acc += A(kd, kh) . B, B = the seven channel words of the lane's cell at cell[0 .. 6] (+ a zero eighth slot)
__device__ __forceinline__ xt_f4 xt_mm3(const xt_u4 ah, const xt_u4 al, const float* __restrict__ cell, xt_f4 acc) {
    xt_u4 bh, bl;
    xt_bfrag(cell, bh, bl);
    return xt_mm3b(ah, al, bh, bl, acc);
}

// One 3x3x3 convolution of the 1-channel block on zero-padded grids [D + 2][H + 2][Wp] (Wp = W + 2 rounded up to an ODD number of words): a thread
// takes S neighbouring cells of a row — per tap row S + 2 LDS reads feed 3 S FMAs (one cell per thread: 27 reads for 27 FMAs, and the index arithmetic
// once per cell) — and consecutive lanes take the same strip of consecutive ROWS: their addresses are an odd pitch apart, every read of a wave
// touches every bank once.  (With consecutive lanes along a row — strips four words apart — the reads were 4-way bank conflicts: half of the kernel's
// LDS cycles, profiles/r04_pt_tower_sq_counters.txt.)  Every cell's sum runs in the order (kd, kh, kw).  RES: out is the INPUT of the block (x) and
// holds x at the cell: out = relu(conv + x), in place.  (A strip may hang over the end of its row: those reads hit the next row or the floats behind
// the grid — inside the allocation — and feed only sums that are dropped.)
// Division of a small non-negative index by a run-time extent (n * d < 2^32): one multiply-high with a reciprocal computed once per kernel — the compiler's
// general u32 division is ~20 vector instructions, and the stages' index arithmetic (three per loaded voxel group, two per strip, four per tile) was a
// quarter of the 1-channel block's instructions.
struct XtFd { uint32_t m, d; };
__device__ __forceinline__ XtFd xt_fd(int d) { return XtFd{d > 1 ? 0xFFFFFFFFu / (uint32_t)d + 1u : 0u, (uint32_t)d}; }
__device__ __forceinline__ int xt_q(int n, XtFd f) { return f.d > 1 ? (int)__umulhi((uint32_t)n, f.m) : n; }

template <int S, bool RES, bool PACK = false>      // PACK: the result is stored as hi << 16 | lo (matrix mode 1: the aligning convolution reads it as bf16 fragments)
__device__ __forceinline__ void xt_conv1_strips(const float* __restrict__ wgt, const float* in, float* out, int D, int H, int W, int tid, int nthr) {
    const int Hp = H + 2, Wp = (W + 2) | 1, rows = D * H, nstrip = rows * ((W + S - 1) / S);
    float wk[27];
#pragma unroll
    for (int k = 0; k < 27; k++) wk[k] = wgt[k];
    const float bias = wgt[27];
    const XtFd frows = xt_fd(rows), fH = xt_fd(H);
    for (int i = tid; i < nstrip; i += nthr) {
        const int sx = xt_q(i, frows), r = i - sx * rows, d = xt_q(r, fH), h = r - d * H, w0 = sx * S;
        const int pi = (d * Hp + h) * Wp + w0;                 // tap (0, 0, 0) of the strip's first cell
        float acc[S];
#pragma unroll
        for (int j = 0; j < S; j++) acc[j] = bias;
#pragma unroll
        for (int kd = 0; kd < 3; kd++)
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float* row = in + pi + (kd * Hp + kh) * Wp;
                float v[S + 2];
#pragma unroll
                for (int j = 0; j < S + 2; j++) v[j] = row[j];
#pragma unroll
                for (int j = 0; j < S; j++) {
                    acc[j] += wk[(kd * 3 + kh) * 3 + 0] * v[j];
                    acc[j] += wk[(kd * 3 + kh) * 3 + 1] * v[j + 1];
                    acc[j] += wk[(kd * 3 + kh) * 3 + 2] * v[j + 2];
                }
            }
        float* o = out + pi + (Hp + 1) * Wp + 1;
#pragma unroll
        for (int j = 0; j < S; j++)
            if (w0 + j < W) { const float v_ = fmaxf(RES ? acc[j] + o[j] : acc[j], 0.f); o[j] = PACK ? xt_pack(v_) : v_; }
    }
}

// The same, four cells per thread as TWO packed sums (v_pk_fma_f32): a tap row's pairs (v0 v1) (v1 v2) (v2 v3) (v3 v4) (v4 v5) are five two-word LDS reads —
// a read lands in an aligned register pair, so the pairs cost no moves — and feed six packed FMAs with the weight on both halves (scalar pairs): 11
// instructions per tap row and four cells where the strips of three take 3 reads + 9 FMAs for three.  Same order of the sums (kd, kh, kw): same bits.
template <bool RES, bool PACK>
__device__ __forceinline__ void xt_conv1_pk4(const float* __restrict__ wgt, const float* in, float* out, int D, int H, int W, int tid, int nthr) {
    const int Hp = H + 2, Wp = (W + 2) | 1, rows = D * H, nstrip = rows * ((W + 3) >> 2);
    xt_f2 wk[27];
#pragma unroll
    for (int k = 0; k < 27; k++) wk[k] = xt_f2{wgt[k], wgt[k]};
    const float bias = wgt[27];
    const XtFd frows = xt_fd(rows), fH = xt_fd(H);
    for (int i = tid; i < nstrip; i += nthr) {
        const int sx = xt_q(i, frows), r = i - sx * rows, d = xt_q(r, fH), h = r - d * H, w0 = sx * 4;
        const int pi = (d * Hp + h) * Wp + w0;                 // tap (0, 0, 0) of the strip's first cell
        xt_f2 a01 = {bias, bias}, a23 = {bias, bias};
#pragma unroll
        for (int kd = 0; kd < 3; kd++)
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float* row = in + pi + (kd * Hp + kh) * Wp;
                const xt_f2 p01 = {row[0], row[1]}, p12 = {row[1], row[2]}, p23 = {row[2], row[3]}, p34 = {row[3], row[4]}, p45 = {row[4], row[5]};
                const int k = (kd * 3 + kh) * 3;
                a01 += wk[k] * p01; a01 += wk[k + 1] * p12; a01 += wk[k + 2] * p23;
                a23 += wk[k] * p23; a23 += wk[k + 1] * p34; a23 += wk[k + 2] * p45;
            }
        float* o = out + pi + (Hp + 1) * Wp + 1;
        const float acc[4] = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (w0 + j < W) { const float v_ = fmaxf(RES ? acc[j] + o[j] : acc[j], 0.f); o[j] = PACK ? xt_pack(v_) : v_; }
    }
}

// Which tiles a wave of the matrix stages takes: T tile groups (one group = the three depth slices of 16 cell pairs: 42 + 63 + 42 instructions)
// dealt round-robin, whole groups — except in a last, partial round: there the waves that would idle take the MIDDLE slice of a group that
// another wave (on another SIMD) owns, which keeps the outer two.  Item k of wave wv: tile group *t, slices *dm (bit d); false: no more items.
// (30 groups over 16 waves: the SIMDs carry 1113 / 1113 / 1092 / 1092 instructions instead of 1176 / 1176 / 1029 / 1029.)
__device__ __forceinline__ bool xt_tile_item(int k, int wv, int nw, int T, int* t, int* dm) {
    const int J = T / nw, R = T - J * nw;
    if (k < J) { *t = k * nw + wv; *dm = 7; return true; }
    if (k > J || R == 0) return false;
    if (wv < R) { *t = J * nw + wv; *dm = (wv + nw - R < nw && wv >= 2 * R - nw) ? 5 : 7; return true; }
    const int v = wv - (nw - R);                       // an idle wave: the middle slice of wave v's group
    if (v < 0) return false;
    *t = J * nw + v; *dm = 2;
    return true;
}

// -DXT_PHASE_TIMING (make ttiming; tools/tower_probe.py with XT_PHASES=1): thread 0's cycle count per stage replaces the first floats of the env's output row
#ifdef XT_PHASE_TIMING
#define XT_LAP(k) do { if (tid == 0) xt_lap[k] = __builtin_readcyclecounter(); } while (0)
#else
#define XT_LAP(k) do { } while (0)
#endif

template <int BT, bool NET, int MM>
__global__ void __launch_bounds__(BT) xr_ob_tower_kernel(const float* __restrict__ head, int64_t stride, int n_envs, XtDims g,
                                                          const float* __restrict__ wt, float* __restrict__ out, int normalize, XnArgs na) {
    extern __shared__ __attribute__((aligned(16))) float xt_smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int e = blockIdx.x;
    if (e >= n_envs) return;
    const int D = g.D, H = g.H, W = g.W, N = D * H * W, HW = H * W;
    const int od = g.od, oh = g.oh, ow = g.ow;
    const int he1 = oh + 3, we1 = ow + 3;            // extent of the block's first activation that the inside cells read
    const int nB = 7 * od * oh * ow;
    float* bufB = xt_smem;                            // [od][oh][ow][7]: the channels of a cell are neighbours
    float* bufC1 = xt_smem + nB;                      // [3][he1][we1][7]
    // The 1-channel block works on ZERO-PADDED grids (a one-voxel halo): a tap is one LDS read at a fixed offset — with bounds checks it was ~10 VALU
    // instructions per FMA.  x sits in the first activation's space and becomes a = block(x) IN PLACE (the residual is the cell's own x, read by the
    // thread that writes a there): the aligning convolution then reads a padded grid too; the intermediate y sits in b's space (dead before b is
    // written) or, where that is too small (narrow grids), behind x.
    const int Hp = H + 2, Wp = (W + 2) | 1, Np = (D + 2) * Hp * Wp;          // (an odd row pitch: see xt_conv1_strips)
    float* xpad = bufC1;
    float* ypad = g.y_in_b ? bufB : bufC1 + Np;

#ifdef XT_PHASE_TIMING
    unsigned long long xt_lap[12];
#endif
    XT_LAP(0);
    const int lane = tid & 63, nw = nthr >> 6, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pn = lane & 15, q = lane >> 4;             // (7-channel block) B operand: cell pair pn, input column offset q; D: rows 4q .. 4q + 3 of pair pn
    const int dwv = q >> 1, co0 = 4 * (q & 1);           // D rows -> output column 2p + dwv, channels co0 .. co0 + 3
    const float kv = wt[XT_KV + lane];                // (what the last stage starts from: fetched here, not behind the last barrier)
    const int zidx = g.tail;                          // seven words that stay 0: what a tap outside the data reads (one per channel)
    float* red = xt_smem + g.tail + 8;                // [nthr / 64][cols + 2][3]: every wave's column sums of the last stage
    for (int i = tid; i < 8 + (nthr >> 6) * (g.cols + 2) * 3; i += nthr) xt_smem[g.tail + i] = 0.f;

    float wA[MM ? 1 : 64];                               // matrix mode 0: the fp32 A operands of a convolution; mode 1: its bf16 fragments
    xt_u4 wAh[MM ? 9 : 1], wAl[MM ? 9 : 1];
    auto load_w = [&](int off_f32, int off_bf16) {
        if constexpr (MM) xt_load_wAb(wt + off_bf16, lane, wAh, wAl);
        else xt_load_wA(wt + off_f32, lane, wA);
    };
#ifdef XT_PHASE_TIMING
    unsigned long long xt_a0 = 0, xt_a1 = 0, xt_a2 = 0;         // every wave: align1's tiles / the next stage's operands arriving
#endif
    if constexpr (NET) {
        // ==== the net tower's SPARSE front end: P = align1(block(x)) of a net's 7 planes without ever forming them (agents.FusedNetTower) ====
        // x is 1 at the net's access points (plane 0; planes 1..6: the aliased "has a same-net axis neighbour" flag) and 0 elsewhere.  With S1 / S2 =
        // the 3x3x3 / 5x5x5 neighbourhoods of the access points:  y = relu(conv_a(x) + b_a) differs from relu(b_a) on S1 only;  o1 = relu(conv_b(y) +
        // b_b + x) differs from bg1 = block(0) on S2 only, where bg1 = relu(zbg[class of the voxel]);  P = align1(bg1) + align1(o1 - bg1): `na.bg`
        // (per shape, from the caller) plus align1's taps of the S2 voxels.
        // Sets are bitmasks, ONE 32-bit word per (d, h) row of the plane (W <= 32), + popcount prefixes: the slot of a voxel = voxels of the set before
        // it — lists in row order whatever order the marks arrive in.  Rows are ordered by (d + 1) % sd first: an S2 voxel reaches an output depth
        // only through the taps kd = (d + 1) mod sd (+ sd ...), so the voxels of one residue class run the same taps — uniform weights, all lanes busy.
        // The taps are SCATTERED from the voxels (every (voxel, tap) pair is real work; gathered per output cell, 87 % of the lanes idled through taps
        // that hit nothing: 275 k of 430 k cycles) with INTEGER atomics in fixed point (order-independent: same input, same bits): scale =
        // 2^30 / (sum over taps and input channels of max_co |W| x max |o1 - bg1|), an upper bound of every partial sum.
        const int HW = H * W, R = D * H;
        const int r_ = na.pair_region[e], a = na.pair_net[e];
        bool bad = (unsigned)r_ >= (unsigned)na.n_regions;
        XrRegionDev Rg = na.regions[bad ? 0 : r_];
        bad = bad || Rg.Z != D || Rg.Y != H || Rg.X != W || a < 1 || a > Rg.n_nets;
        if (bad) { if (tid == 0) na.flags[e] = 2; return; }
        const int lo = na.net_csr[Rg.net_off + a], nap = na.net_csr[Rg.net_off + a + 1] - lo;      // (<= XR_MAX_AP_PER_NET = 128, xr_batch_load_regions)
        uint32_t* msk = reinterpret_cast<uint32_t*>(bufC1);                 // [4][R]: S1, S2, access points, their "adjacent" flags
        int* pre = reinterpret_cast<int*>(bufC1) + 4 * R;                    // [2][R]: set voxels before row i of S1 / S2
        int* s_ap = pre + 2 * R;                                             // [128][2]: the access point's d << 11 | h << 5 | w in the plane, its "adjacent" flag
        int* s_cnt = s_ap + 256;                                             // n1, n2, bits of max |dq|
        int* s_dp = s_cnt + 8;                                               // dperm[32] (d -> position in class order), dinv[32], cs[sd + 1] (first position of every class)
        float* s_tab = reinterpret_cast<float*>(s_dp + 112);                 // A0[216] AS[216] ba[8] zbg[512]
        const int fixed = 6 * R + 256 + 8 + 112 + 952;
        const int sd = g.sd;
        for (int i = tid; i < 4 * R; i += nthr) msk[i] = 0u;
        for (int i = tid; i < 952; i += nthr) s_tab[i] = i < 440 ? wt[XN_A0 + i] : wt[XN_ZBG + i - 440];
        if (tid < nap) {                                                     // (integer divisions by run-time values cost ~40 instructions each: once per access point)
            const int v = na.ap_feat[Rg.ap_off + lo + tid], f = v & 0x7FFFFFFF, d = f / HW, rem = f - d * HW, h = rem / W;
            s_ap[2 * tid] = (d << 11) | (h << 5) | (rem - h * W); s_ap[2 * tid + 1] = (int)((uint32_t)v >> 31);
        }
        if (tid == 0) {
            int pos = 0;
            for (int c = 0; c < sd; c++) {
                s_dp[64 + c] = pos;
                for (int d = 0; d < D; d++)
                    if ((d + 1) % sd == c) { s_dp[d] = pos; s_dp[32 + pos] = d; pos++; }
            }
            s_dp[64 + sd] = pos;
            s_cnt[2] = 0; s_cnt[3] = 0;
        }
        __syncthreads();
        for (int i = tid; i < nap * 25; i += nthr) {
            const int k = i / 25, o = i - k * 25, dd = o / 5 - 2, dh = o % 5 - 2;
            const int f = s_ap[2 * k], d = (f >> 11) + dd, h = ((f >> 5) & 63) + dh, w = f & 31;
            if ((unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H) {
                const int row = s_dp[d] * H + h;
                const int l2 = max(w - 2, 0), h2 = min(w + 2, W - 1);
                atomicOr(&msk[R + row], ((2u << h2) - 1u) & ~((1u << l2) - 1u));
                if (abs(dd) <= 1 && abs(dh) <= 1) {
                    const int l1 = max(w - 1, 0), h1 = min(w + 1, W - 1);
                    atomicOr(&msk[row], ((2u << h1) - 1u) & ~((1u << l1) - 1u));
                }
                if (o == 12) {                                               // the access point itself
                    atomicOr(&msk[2 * R + row], 1u << w);
                    if (s_ap[2 * k + 1]) atomicOr(&msk[3 * R + row], 1u << w);
                }
            }
        }
        __syncthreads();
        if (wv < 2) {                                                        // wave 0: S1, wave 1: S2 — exclusive popcount prefixes
            const uint32_t* m = msk + wv * R;
            int* pr = pre + wv * R;
            const int per = (R + 63) >> 6, base = lane * per;
            int sum = 0;
            for (int j = 0; j < per; j++) sum += base + j < R ? __popc(m[base + j]) : 0;
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            int run = incl - sum;
            for (int j = 0; j < per; j++)
                if (base + j < R) { pr[base + j] = run; run += __popc(m[base + j]); }
            if (lane == 63) s_cnt[wv] = incl;
        }
        __syncthreads();
        const int n1 = s_cnt[0], n2 = s_cnt[1];
        // lists: S1 (voxel ids + dy) in b's space — free until the scatter —, S2 (voxel ids + dq) behind the fixed part
        const int c1_words = g.tail - nB;
        if ((n1 + 1) / 2 + 7 * n1 > nB || fixed + (n2 + 1) / 2 + 7 * n2 > c1_words) { if (tid == 0) na.flags[e] = 1; return; }
        uint16_t* vox1 = reinterpret_cast<uint16_t*>(bufB);                  // d << 11 | h << 5 | w
        float* dy = bufB + (n1 + 1) / 2;
        uint16_t* vox2 = reinterpret_cast<uint16_t*>(bufC1 + fixed);
        float* dq = bufC1 + fixed + (n2 + 1) / 2;
        for (int t = tid; t < 2 * R; t += nthr) {
            const int which = t >= R, wd = t - which * R;
            uint32_t bits = msk[which * R + wd];
            if (!bits) continue;
            int p = pre[which * R + wd];
            uint16_t* vx = which ? vox2 : vox1;
            const int dp = wd / H, dh = (s_dp[32 + dp] << 11) | ((wd - dp * H) << 5);
            while (bits) { const int b = __builtin_ctz(bits); vx[p++] = (uint16_t)(dh | b); bits &= bits - 1; }
        }
        __syncthreads();
        XT_LAP(1);
        // ---- dy = relu(b_a + the access points' stamps) - relu(b_a) on S1
        float ymax = 0.f;
        for (int s = tid; s < n1; s += nthr) {
            const int ev = vox1[s], d = ev >> 11, h = (ev >> 5) & 63, w = ev & 31;
            float acc[7];
#pragma unroll
            for (int c = 0; c < 7; c++) acc[c] = s_tab[432 + c];
            for (int k = 0; k < nap; k++) {
                const int f = s_ap[2 * k], kd = (f >> 11) - d + 1, kh = ((f >> 5) & 63) - h + 1, kw = (f & 31) - w + 1;
                if ((unsigned)kd < 3u && (unsigned)kh < 3u && (unsigned)kw < 3u) {
                    const int t = ((kd * 3 + kh) * 3 + kw) * 8;
                    const bool adj = s_ap[2 * k + 1] != 0;
#pragma unroll
                    for (int c = 0; c < 7; c++) acc[c] = acc[c] + s_tab[t + c] + (adj ? s_tab[216 + t + c] : 0.f);
                }
            }
            float ym = 0.f;
#pragma unroll
            for (int c = 0; c < 7; c++) { const float y_ = fmaxf(acc[c], 0.f) - fmaxf(s_tab[432 + c], 0.f); dy[s * 7 + c] = y_; ym = fmaxf(ym, fabsf(y_)); }
            ymax = fmaxf(ymax, ym);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        if (lane == 0 && ymax > 0.f) atomicMax(&s_cnt[3], __float_as_int(ymax));            // (non-negative floats order like their bit patterns)
        int* zacc = reinterpret_cast<int*>(dq);                                              // conv_b's sums over S1, fixed point, where dq will be
        for (int i = tid; i < 7 * n2; i += nthr) zacc[i] = 0;
        __syncthreads();
        XT_LAP(2);
        // ---- dq = o1 - bg1 on S2.  conv_b(dy) is SCATTERED from the S1 voxels (every neighbour of an S1 voxel inside the grid is an S2 voxel: each (voxel, tap)
        // pair is real work — gathered per S2 voxel, three of four neighbour probes found nothing: 22 k cycles against 6 k) with integer atomics in fixed point,
        // scale = 2^30 / (sum over taps and input channels of max_co |W_b| x max |dy|); work units = the 27 taps, one tap's weights per wave at a time
        const float ybound = wt[XN_WSUM + 1] * __int_as_float(s_cnt[3]);
        if (ybound > 0.f) {
            const float yscale = 1073741824.f / ybound;
            const int* pre2 = pre + R;
            for (int t = wv; t < 27; t += nw) {
                const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
                const xt_f2* __restrict__ wk = reinterpret_cast<const xt_f2*>(wt + XN_WB + t * 56);
                xt_f2 wr[28];
#pragma unroll
                for (int i = 0; i < 28; i++) wr[i] = wk[i];
                for (int s = lane; s < n1; s += 64) {
                    const int ev = vox1[s], vd = (ev >> 11) - kd + 1, vh = ((ev >> 5) & 63) - kh + 1, vw = (ev & 31) - kw + 1;      // z[v] takes W_b[tap k] . dy[v + k - 1]
                    if ((unsigned)vd >= (unsigned)D || (unsigned)vh >= (unsigned)H || (unsigned)vw >= (unsigned)W) continue;
                    const int vrow = s_dp[vd] * H + vh;
                    const int slot = pre2[vrow] + __popc(msk[R + vrow] & ((1u << vw) - 1u));
                    const float* yr = dy + s * 7;
                    xt_f2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                    for (int ci = 0; ci < 7; ci++) {
                        const float yv = yr[ci];
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j] += wr[ci * 4 + j] * yv;
                    }
#pragma unroll
                    for (int co = 0; co < 7; co++) atomicAdd(zacc + slot * 7 + co, __float2int_rn(acc[co >> 1][co & 1] * yscale));
                }
            }
        }
        __syncthreads();
        float dmax = 0.f;
        {
            const float yinv = ybound * (1.f / 1073741824.f);
            for (int s = tid; s < n2; s += nthr) {
                const int ev = vox2[s], d = ev >> 11, h = (ev >> 5) & 63, w = ev & 31, row = s_dp[d] * H + h;
                const int cls = ((((d > 0) | ((d < D - 1) << 1)) << 4) | (((h > 0) | ((h < H - 1) << 1)) << 2) | ((w > 0) | ((w < W - 1) << 1))) * 8;
                const bool isap = (msk[2 * R + row] >> w) & 1u, isadj = (msk[3 * R + row] >> w) & 1u;
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    const float zb = s_tab[440 + cls + c], xv = c == 0 ? (isap ? 1.f : 0.f) : (isadj ? 1.f : 0.f);
                    const float q_ = fmaxf(zb + (float)zacc[s * 7 + c] * yinv + xv, 0.f) - fmaxf(zb, 0.f);
                    dq[s * 7 + c] = q_;
                    dmax = fmaxf(dmax, fabsf(q_));
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
        if (lane == 0 && dmax > 0.f) atomicMax(&s_cnt[2], __float_as_int(dmax));
        __syncthreads();
        XT_LAP(3);
        // ---- the scatter: fixed-point sums of align1's taps in b's space (dy is dead), then P = background + sums
        int* accB = reinterpret_cast<int*>(bufB);
        for (int i = tid; i < nB; i += nthr) accB[i] = 0;
        load_w(XT_C1, XT_C1B);                                 // (the first matrix stage's operands: in flight under the scatter)
        __syncthreads();
        XT_LAP(11);
        const float qmax = __int_as_float(s_cnt[2]);
        const float bound = wt[XN_WSUM] * qmax;
        const float scale = qmax > 0.f ? 1073741824.f / bound : 0.f;
        if (qmax > 0.f) {
            // work units (wave-uniform): the 25 tap rows (kd, kh), dealt round-robin over the waves; a unit runs its 5 taps over every voxel of the
            // residue class kd belongs to (chunks of 64): the 56 weights of a tap are fetched once per wave and tap (scalar loads), not once per chunk
            const int* pre2 = pre + R;
            const int inv_sd = 65536 / sd + 1;
            for (int u = wv; u < 25; u += nw) {
                const int kd = u / 5, kh = u - kd * 5, c = kd % sd;
                const int r0 = s_dp[64 + c] * H, r1 = s_dp[64 + c + 1] * H;
                const int s_lo = r0 < R ? pre2[r0] : n2, s_hi = r1 < R ? pre2[r1] : n2;
                if (s_hi <= s_lo) continue;
#pragma unroll 1
                for (int kw = 0; kw < 5; kw++) {
                    const xt_f2* __restrict__ wk = reinterpret_cast<const xt_f2*>(wt + XN_WL + ((kd * 5 + kh) * 5 + kw) * 56);
                    xt_f2 wr[28];
#pragma unroll
                    for (int i = 0; i < 28; i++) wr[i] = wk[i];
                    for (int s = s_lo + lane; s < s_hi; s += 64) {
                        const int ev = vox2[s], d = ev >> 11, h = (ev >> 5) & 63, w = ev & 31;
                        const int th = h + 1 - kh, td = d + 1 - kd, tw = w + 1 - kw;           // (td: a multiple of sd — that is what the class is; sh = sw = 1: the launcher's condition)
                        const int dz = (td * inv_sd) >> 16;                                     // td / sd without the division (td < 64)
                        if ((unsigned)th >= (unsigned)oh || td < 0 || dz >= od || (unsigned)tw >= (unsigned)ow) continue;
                        const float* qr = dq + s * 7;
                        xt_f2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                        for (int ci = 0; ci < 7; ci++) {
                            const float qv = qr[ci];
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[j] += wr[ci * 4 + j] * qv;
                        }
                        int* o = accB + ((dz * oh + th) * ow + tw) * 7;
#pragma unroll
#ifdef XN_PLAIN_RMW      /* timing experiment only (racy): what the scatter costs without the atomics */
                        for (int co = 0; co < 7; co++) o[co] += __float2int_rn(acc[co >> 1][co & 1] * scale);
#else
                        for (int co = 0; co < 7; co++) atomicAdd(o + co, __float2int_rn(acc[co >> 1][co & 1] * scale));
#endif
                    }
                }
            }
        }
        __syncthreads();
        {
            const float inv = qmax > 0.f ? bound * (1.f / 1073741824.f) : 0.f;
            if constexpr (MM != 0) {
                for (int c = tid; c * 7 < nB; c += nthr) {                                // (a thread reads its cell's seven sums, then writes the cell's seven words over them)
                    float v[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) v[k] = na.bg[c * 7 + k] + (float)accB[c * 7 + k] * inv;
                    xt_store_cell(bufB + c * 7, v);
                }
            } else {
                for (int i = tid; i < nB; i += nthr) bufB[i] = na.bg[i] + (float)accB[i] * inv;
            }
        }
        __syncthreads();
    } else {
    const float* __restrict__ src = head + (int64_t)e * stride;
    // the env's plane is asked for BEFORE the buffers are zeroed (three 16-byte loads per thread cover 12 K voxels): the fill hides the fetch
    constexpr int PF = 3;
    xt_f4 pfv[PF];
    const XtFd fW = xt_fd(W), fH = xt_fd(H);
    if (g.vec_load) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i4 = tid + k * nthr;
            pfv[k] = i4 < (N >> 2) ? reinterpret_cast<const xt_f4*>(src)[i4] : xt_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int i = tid; i < Np + 8; i += nthr) { xpad[i] = 0.f; if (i < Np) ypad[i] = 0.f; }      // (+ 8 zero words behind x: the matrix form of align1 reads one word past a row)
    __syncthreads();
    if (g.vec_load) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i4 = tid + k * nthr;
            if (i4 < (N >> 2)) {
                const int i = i4 << 2, r = xt_q(i, fW), w = i - r * W, d = xt_q(r, fH), h = r - d * H;
                float* o = xpad + ((d + 1) * Hp + h + 1) * Wp + w + 1;
                o[0] = pfv[k][0]; o[1] = pfv[k][1]; o[2] = pfv[k][2]; o[3] = pfv[k][3];
            }
        }
        for (int i4 = tid + PF * nthr; i4 < (N >> 2); i4 += nthr) {
            const int i = i4 << 2, w = i % W, r = i / W, h = r % H, d = r / H;
            const xt_f4 v = reinterpret_cast<const xt_f4*>(src)[i4];
            float* o = xpad + ((d + 1) * Hp + h + 1) * Wp + w + 1;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        }
    } else {
        for (int i = tid; i < N; i += nthr) {
            const int w = i % W, h = (i / W) % H, d = i / HW;
            xpad[((d + 1) * Hp + h + 1) * Wp + w + 1] = src[i];
        }
    }
    __syncthreads();
    XT_LAP(1);
    // ---- ResidualBlock(1): y = relu(conv3(x) + b1), then a = relu(conv3(y) + b2 + x) over x ------------------------------------------------
    if (g.strip == 4) xt_conv1_pk4<false, false>(wt + XT_A1, xpad, ypad, D, H, W, tid, nthr);
    else if (g.strip == 5) xt_conv1_strips<5, false>(wt + XT_A1, xpad, ypad, D, H, W, tid, nthr);
    else xt_conv1_strips<3, false>(wt + XT_A1, xpad, ypad, D, H, W, tid, nthr);
    __syncthreads();
    XT_LAP(2);
    const bool al1_mm = MM && g.sh == 1 && g.sw == 1;      // align1 on the matrix pipe too (matrix mode 1, unit strides in h and w)
    if (g.strip == 4) {
        if (al1_mm) xt_conv1_pk4<true, MM != 0>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
        else xt_conv1_pk4<true, false>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
    } else if (al1_mm) {
        if (g.strip == 5) xt_conv1_strips<5, true, MM != 0>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
        else xt_conv1_strips<3, true, MM != 0>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
    } else {
        if (g.strip == 5) xt_conv1_strips<5, true>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
        else xt_conv1_strips<3, true>(wt + XT_A2, ypad, xpad, D, H, W, tid, nthr);
    }
    __syncthreads();
    XT_LAP(3);
    // the A operand of the 7-channel block's first convolution: 16 loads per lane, issued here so that the aligning convolution hides them
    // (it fetches nothing through the vector memory path; earlier, the 64 registers would squeeze the 1-channel stages)
    if constexpr (MM == 0) load_w(XT_C1, XT_C1B);      // (matrix mode 1: after align1 — its six accumulators and fragments need the registers)
    // ---- align1: 5x5x5, 1 -> 7 channels, stride (sd,sh,sw), padding 1 ------------------------------------------------------------
    const int ncellB = od * oh * ow;
#ifdef XT_PHASE_TIMING
    xt_a0 = __builtin_readcyclecounter();
#endif
    if constexpr (MM != 0) {
      if (al1_mm) {
        // ---- align1 on v_mfma_f32_16x16x32_bf16, split operands.  Rows (dw, co): two output columns x 8 channel slots; columns: 16 cell pairs (hz, p) of one
        // output slice.  The B operand is the RAW packed word of an input cell (x_hi << 16 | x_lo as the block's second convolution stored it): K slot 2j is x_lo
        // and 2j + 1 is x_hi of cell j, so an instruction covers 16 cells and the lanes permute nothing — fragment kind 0 holds w_hi on both halves
        // (w_hi (x_hi + x_lo)), kind 1 holds w_lo on the x_hi half: two instructions per 16 cells for the same three products.  Cells c = 6 kh + c6: the 5 kernel rows
        // x the 6 input columns 2p - 1 .. 2p + 4 of one kd (c6 = dw + kw; 30 of 32 cells, weight 0 elsewhere); step (kd, g): cells 16 g .. + 15, the lane's quarter
        // holds cells 16 g + 4 q .. + 3 = two adjacent pairs of words.  Per 32 output cells and kd: 4 instructions, 4 two-word reads, no vector work beside addresses.
        // (forms before, same products: [4 kernel rows x 8 columns] x 10 steps and [5 rows x 6 columns] x 5 steps with hi / lo fragments permuted out of the packed
        // words — 8 reads + 8 v_perm per 3 instructions: 30.0 k / 29.3 k cycles, the vector form 32.9 k.)
        const xt_u4* __restrict__ wsrc = reinterpret_cast<const xt_u4*>(wt + XT_AL1B) + lane;
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = wt[XT_AL1 + 1000 + co0 + i];
        int offs[4];                                                                    // the lane's pairs (g, half): word offset from (kernel row 0, column 2p - 1)
#pragma unroll
        for (int i = 0; i < 4; i++) { const int c = 16 * (i >> 1) + 4 * q + 2 * (i & 1), c2 = c < 30 ? c : 0; offs[i] = (c2 / 6) * Wp + c2 % 6; }
        const int ppr = (ow + 1) >> 1, npair = oh * ppr, T = (npair + 15) >> 4, ntile = od * T;
        const XtFd fT = xt_fd(T), fppr = xt_fd(ppr);
        // a wave takes its tiles (wv, wv + nw, ...) up to SIX at a time with the step loop outside: the fragments of a step are fetched once per wave and round
        // (a step ahead: six tiles of work cover the fetch), and the tiles of a round have no branch between them — their reads and matrix instructions interleave
        // (with a wave-uniform `break` per tile the tiles ran one after the other, each waiting for its own LDS reads: a wave alone took 13 k cycles for five tiles)
        auto al1_round = [&](auto nt_c, const int t0) {
            constexpr int NT = decltype(nt_c)::value;
            xt_f4 acc[NT];
            int rb[NT];
            bool ok[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int tile = t0 + j * nw, dz = xt_q(tile, fT), f = (tile - dz * T) * 16 + pn;
                ok[j] = f < npair;
                const int fc = ok[j] ? f : npair - 1, hz = xt_q(fc, fppr), p = fc - hz * ppr;
                rb[j] = (dz * g.sd * Hp + hz) * Wp + 2 * p;                           // kernel row 0 of slice kd = 0, input column 2p - 1 (the grid's halo is the padding)
                acc[j] = xt_f4{bias[0], bias[1], bias[2], bias[3]};
            }
            xt_u4 a00 = wsrc[0], a01 = wsrc[64], a10 = wsrc[128], a11 = wsrc[192];
#pragma unroll 1
            for (int kd = 0; kd < 5; kd++) {
                const xt_u4* __restrict__ wn = wsrc + min(kd + 1, 4) * 256;
                const xt_u4 n00 = wn[0], n01 = wn[64], n10 = wn[128], n11 = wn[192];
                const uint32_t* __restrict__ slab = reinterpret_cast<const uint32_t*>(xpad) + kd * Hp * Wp;
#pragma unroll
                for (int gk = 0; gk < 2; gk++) {                                        // cells 16 gk .. + 15: the round's reads, then its instructions
                    xt_u4 b[NT];
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const uint32_t* __restrict__ pw = slab + rb[j];
                        b[j] = xt_u4{pw[offs[2 * gk]], pw[offs[2 * gk] + 1], pw[offs[2 * gk + 1]], pw[offs[2 * gk + 1] + 1]};
                    }
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xt_bf8, gk ? a10 : a00), __builtin_bit_cast(xt_bf8, b[j]), acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xt_bf8, gk ? a11 : a01), __builtin_bit_cast(xt_bf8, b[j]), acc[j], 0, 0, 0);
                }
                a00 = n00; a01 = n01; a10 = n10; a11 = n11;
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int tile = t0 + j * nw;
                const int dz = xt_q(tile, fT), f = (tile - dz * T) * 16 + pn, hz = xt_q(f, fppr), p = f - hz * ppr, w = 2 * p + dwv;
                if (ok[j] && w < ow) xt_store_half(bufB + ((dz * oh + hz) * ow + w) * 7, co0 >> 2, acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            }
        };
        constexpr int TC = XT_AL1_TC;
        for (int t0 = wv; t0 < ntile; t0 += nw * TC) {
            const int nt = min(TC, (ntile - t0 + nw - 1) / nw);                         // (wave-uniform)
            if (nt >= 6) al1_round(xt_ic<(TC >= 6 ? 6 : 1)>{}, t0);
            else if (nt == 5) al1_round(xt_ic<(TC >= 5 ? 5 : 1)>{}, t0);
            else if (nt == 4) al1_round(xt_ic<(TC >= 4 ? 4 : 1)>{}, t0);
            else if (nt == 3) al1_round(xt_ic<(TC >= 3 ? 3 : 1)>{}, t0);
            else if (nt == 2) al1_round(xt_ic<(TC >= 2 ? 2 : 1)>{}, t0);
            else al1_round(xt_ic<1>{}, t0);
        }
      }
    }
    // (consecutive lanes take consecutive ROWS of one column: their reads are an odd pitch apart.  Measured and dropped: a work item of HALF a cell —
    //  four of the eight channel slots, 80 chunks dealt evenly over 16 waves instead of 2.45 passes rounded up to 3: 47.6 k cycles against 32.9 k,
    //  the LDS read and the scalar weight load per tap do not shrink with the channels)
    for (int q = (MM != 0 && al1_mm) ? ncellB : tid; q < ncellB; q += nthr) {
        const int hz = q % oh, wz = (q / oh) % ow, dz = q / (ow * oh), i = (dz * oh + hz) * ow + wz;
        const float* ap = xpad + (dz * g.sd * Hp + hz * g.sh) * Wp + wz * g.sw;          // tap (0, 0, 0): the grid's halo is the convolution's padding
        xt_f2 acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = reinterpret_cast<const xt_f2*>(wt + XT_AL1 + 1000)[k];
#pragma unroll 1
        for (int kd = 0; kd < 5; kd++) {
#pragma unroll 1
            for (int kh = 0; kh < 5; kh++) {
                const float* row = ap + (kd * Hp + kh) * Wp;
#pragma unroll
                for (int kw = 0; kw < 5; kw++) {
                    const float v = row[kw];
                    const xt_f2* __restrict__ wk = reinterpret_cast<const xt_f2*>(wt + XT_AL1 + ((kd * 5 + kh) * 5 + kw) * 8);
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[k] += wk[k] * v;
                }
            }
        }
        if constexpr (MM != 0) {
            const float v[7] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0]};
            xt_store_cell(bufB + i * 7, v);
        } else {
#pragma unroll
            for (int co = 0; co < 7; co++) bufB[i * 7 + co] = acc[co >> 1][co & 1];
        }
    }
#ifdef XT_PHASE_TIMING
    xt_a1 = __builtin_readcyclecounter();
#endif
    if constexpr (MM != 0) load_w(XT_C1, XT_C1B);
#ifdef XT_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xt_a2 = __builtin_readcyclecounter();
#endif
    __syncthreads();
    }       // (!NET)
    XT_LAP(4);
    // ---- ResidualBlock(7) on the standard grid [7,3,64,64] (P = b, zero elsewhere): first activation on h < oh + 3, w < ow + 3 ----
    // Both 7 -> 7-channel 3x3x3 convolutions run on the matrix pipe as an implicit GEMM of v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered
    // fma chain).  With 7 output channels a plain mapping fills 7 of the 16 rows; here one instruction computes TWO neighbouring output
    // columns of 16 cell pairs:
    //   M (A, rows)  = (dw, co): output column 2p + dw, channel co (8 slots, 7 used)                          -> 14 of 16 rows
    //   N (B, cols)  = 16 consecutive cell pairs (h, p) of one depth slice, flattened over the rows
    //   K            = the four input columns 2p - 1 .. 2p + 2 of ONE input channel at ONE (kd, kh): c4 = dw + kw, weight 0 where kw falls outside
    // => 63 instructions (9 (kd, kh) x 7 channels) per 32 cells instead of 189 LDS reads + 756 packed FMAs per cell-thread; the weights
    // sit in 63 VGPRs of every lane for the whole convolution (the scalar-load form spilled its weight SGPRs to VGPR lanes and waited for
    // every s_load: 3.5x its FMA bound), the activations are one LDS read per instruction.  Lanes whose tap falls outside the data read a
    // zero word instead (a zero WEIGHT would not do: 0 x garbage may be NaN).
    // Only the cells with h <= oh and w <= ow see any data (their taps reach h - 1 < oh, w - 1 < ow); every other stored cell is relu(bias).
    // (the fringe of a slice: its last two rows and the last two columns of the rows above — 2 we1 + 2 (oh + 1) cells, one pass)
    const int nfr = 2 * we1 + 2 * (oh + 1);
    for (int i = tid; i < 3 * nfr; i += nthr) {
        const int d = i / nfr, r = i - d * nfr, r2 = r - 2 * we1;
        const int h = r2 < 0 ? oh + 1 + r / we1 : r2 >> 1, w = r2 < 0 ? r % we1 : ow + 1 + (r2 & 1);
        float* o = bufC1 + ((d * he1 + h) * we1 + w) * 7;
        float v[7];
#pragma unroll
        for (int co = 0; co < 7; co++) v[co] = fmaxf(wt[XT_C1 + 4096 + co], 0.f);
        if constexpr (MM != 0) xt_store_cell(o, v);
        else {
#pragma unroll
            for (int co = 0; co < 7; co++) o[co] = v[co];
        }
    }
    const int cbase = (int)(bufC1 - xt_smem);
#ifdef XT_PHASE_TIMING
    const unsigned long long xt_w0 = __builtin_readcyclecounter();      // every wave: when it starts / ends its tiles of the first matrix stage
#endif
    XT_LAP(8);                                         // (timing build: wave 0 after the fill loop / after its own tiles / at the barrier)
    // A wave takes ALL THREE depth slices of its tiles: the lane geometry (two integer divisions, nine tap-row addresses) is computed once per
    // three tiles and the addresses step from slice to slice by one add each — vector instructions and matrix instructions do not overlap
    // on a SIMD (stage time = 32 cycles per matrix instruction + 4 per vector instruction, measured), so every one saved counts.
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = wt[XT_C1 + 4096 + co0 + i];
        const int hc = oh + 1, wc = ow + 1, ppr = (wc + 1) >> 1, npair = hc * ppr, T = (npair + 15) >> 4;
        const int slice = oh * ow * 7;
        const XtFd fppr = xt_fd(ppr);
        for (int item = 0, t = 0, dm = 0; xt_tile_item(item, wv, nw, T, &t, &dm); item++) {
            const int f = t * 16 + pn;
            const bool lv = f < npair;
            const int fc = lv ? f : npair - 1, h = xt_q(fc, fppr), p = fc - h * ppr;
            const int col = 2 * p + q - 1;
            const bool cv = lv && (unsigned)col < (unsigned)ow;
            // the three kh tap rows of this lane in slice 0: float index of channel 0 (the channels of a cell are neighbours: immediate offsets),
            // or the zero words where the row or the column falls outside; ds: what moves a row to the next slice (0 for the zero words)
            const int base0 = (h * ow + col) * 7;
            int am[3], ds[3];
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const bool m = cv && (unsigned)(h + kh - 1) < (unsigned)oh;
                ds[kh] = m ? slice : 0;
                am[kh] = m ? base0 + (kh - 1) * ow * 7 : zidx;
            }
            const int w = 2 * p + dwv;
            const bool st = lv && w < wc;
            float* o = bufC1 + (h * we1 + w) * 7 + co0;
            if constexpr (MM != 0) {
                // matrix mode 1: input-slice-major — the fragments of an input cell row (7 LDS reads + 8 permutes) are built ONCE and feed the up to three output
                // slices it is a tap of (output-major they were built three times: the vector work of the stage, 13 k of its 24 k cycles)
                // The slices of the item (dm: 7 = all three, 5 / 2 = the shared group of a partial round) are a COMPILE-TIME constant of the body: tested at run time
                // (wave-uniform) every triple of matrix instructions sat in its own basic block behind a branch — 27 branches per item, nothing scheduled across them.
                auto body = [&](auto dm_c) {
                    constexpr int DM = decltype(dm_c)::value;
                    xt_f4 acc3[3];
#pragma unroll
                    for (int d = 0; d < 3; d++) acc3[d] = xt_f4{bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
                    for (int dd = 0; dd < 3; dd++) {
                        if (dd >= od) break;                              // (standard padding behind the data's slices) — wave-uniform
#pragma unroll
                        for (int kh = 0; kh < 3; kh++) {
                            __builtin_amdgcn_sched_barrier(0);            // (one cell row's fragments at a time: hoisting all nine rows' reads spills)
                            xt_u4 bh, bl;
                            xt_bfrag(xt_smem + am[kh] + dd * ds[kh], bh, bl);
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                const int kd = dd - d + 1;
                                if (kd < 0 || kd > 2 || !((DM >> d) & 1)) continue;       // (compile time)
                                acc3[d] = xt_mm3b(wAh[kd * 3 + kh], wAl[kd * 3 + kh], bh, bl, acc3[d]);
                            }
                        }
                    }
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        if (!((DM >> d) & 1) || !st) continue;
                        xt_store_half(o - co0 + d * he1 * we1 * 7, co0 >> 2, fmaxf(acc3[d][0], 0.f), fmaxf(acc3[d][1], 0.f), fmaxf(acc3[d][2], 0.f), fmaxf(acc3[d][3], 0.f));
                    }
                };
                if (dm == 7) body(xt_ic<7>{}); else if (dm == 5) body(xt_ic<5>{}); else body(xt_ic<2>{});          // (xt_tile_item deals 7, 5 and 2 only)
                continue;
            }
#pragma unroll 1
            for (int d = 0; d < 3; d++) {
                if (!((dm >> d) & 1)) {                               // another wave's slice (wave-uniform)
#pragma unroll
                    for (int kh = 0; kh < 3; kh++) am[kh] += ds[kh];
                    continue;
                }
                xt_f4 acc = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
                for (int kd = 0; kd < 3; kd++) {
                    const int dd = d + kd - 1;
                    if ((unsigned)dd >= (unsigned)od) continue;       // (d >= od: standard padding; d < 0 or >= 3: the convolution's own) — wave-uniform
#pragma unroll
                    for (int kh = 0; kh < 3; kh++) {
                        const int a = kd == 0 ? am[kh] - ds[kh] : kd == 2 ? am[kh] + ds[kh] : am[kh];
                        if constexpr (MM != 0) {
                            acc = xt_mm3(wAh[kd * 3 + kh], wAl[kd * 3 + kh], xt_smem + a, acc);
                        } else {
#pragma unroll
                            for (int ci = 0; ci < 7; ci++)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[(kd * 3 + kh) * 7 + ci], xt_smem[a + ci], acc, 0, 0, 0);
                        }
                    }
                }
                if (st) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (co0 + i < 7) o[d * he1 * we1 * 7 + i] = fmaxf(acc[i], 0.f);          // (fp32 mode: matrix mode 1 left above)
                }
#pragma unroll
                for (int kh = 0; kh < 3; kh++) am[kh] += ds[kh];
            }
        }
    }
#ifdef XT_PHASE_TIMING
    const unsigned long long xt_w1 = __builtin_readcyclecounter();
#endif
    XT_LAP(9);
    load_w(XT_C2, XT_C2B);                            // (before the barrier: a wave that is done fetches while the others finish)
#ifdef XT_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XT_LAP(10);
#endif
    __syncthreads();
    XT_LAP(5);
    // ---- second convolution + residual + relu on the inside cells (h < oh + 2, w < ow + 2), folded straight into align2's sums ----
    // Same implicit GEMM; the epilogue multiplies every finished cell with its three align2 weights (what it gives to out[w + 1], out[w],
    // out[w - 1]) and adds the products to the wave's own column sums in LDS (no atomics, a fixed order: same input, same bits).
    const int cols = g.cols;
    float* redw = red + (wv * (cols + 2) + 1) * 3;          // (a zero column on either side: the last stage reads without conditions)
    {
        float bias[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bias[i] = wt[XT_C2 + 4096 + co0 + i];
        const int ppr = (cols + 1) >> 1, npair = (oh + 2) * ppr, T = (npair + 15) >> 4;
        const int slice = he1 * we1 * 7;
        const XtFd fppr = xt_fd(ppr);
        for (int item = 0, t = 0, dm = 0; xt_tile_item(item, wv, nw, T, &t, &dm); item++) {
            const int f = t * 16 + pn;
            const bool lv = f < npair;
            const int fc = lv ? f : npair - 1, h = xt_q(fc, fppr), p = fc - h * ppr;
            const int col = 2 * p + q - 1;
            const bool cv = lv && (unsigned)col < (unsigned)we1;
            const int base0 = cbase + (h * we1 + col) * 7;
            int am[3], ds[3];
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const bool m = cv && h + kh - 1 >= 0;                  // (h + kh - 1 <= oh + 2 < he1: always stored)
                ds[kh] = m ? slice : 0;
                am[kh] = m ? base0 + (kh - 1) * we1 * 7 : zidx;
            }
            const int w = 2 * p + dwv;
            const bool ov = lv && w < cols;
            const bool inhw = ov && h < oh && w < ow;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            constexpr bool RB = xt_reuse_b2<NET, MM, BT>();
            constexpr bool PRE = RB && !NET;                          // (768 threads: 170 registers — room to ask for all three slices' epilogue operands up front)
            xt_f4 acc3[RB ? 3 : 1];
            if constexpr (PRE) {
                xt_f4 wkp[3][3];
                float pvp[3][4];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    if (!((dm >> d) & 1)) continue;
                    const xt_f4* __restrict__ wk = reinterpret_cast<const xt_f4*>(wt + XT_AL2 + ((d * 64 + h) * 8 + co0) * 3);
                    wkp[d][0] = wk[0]; wkp[d][1] = wk[1]; wkp[d][2] = wk[2];
                    xt_load_half((inhw && d < od) ? bufB + ((d * oh + h) * ow + w) * 7 : xt_smem + zidx, co0 >> 2, pvp[d]);
                }
#pragma unroll
                for (int d = 0; d < 3; d++) acc3[d] = xt_f4{bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
                for (int dd = 0; dd < 3; dd++) {
#pragma unroll
                    for (int kh = 0; kh < 3; kh++) {
                        __builtin_amdgcn_sched_barrier(0);
                        xt_u4 bh, bl;
                        xt_bfrag(xt_smem + am[kh] + dd * ds[kh], bh, bl);
#pragma unroll
                        for (int d = 0; d < 3; d++) {
                            const int kd = dd - d + 1;
                            if (kd < 0 || kd > 2 || !((dm >> d) & 1)) continue;
                            acc3[d] = xt_mm3b(wAh[kd * 3 + kh], wAl[kd * 3 + kh], bh, bl, acc3[d]);
                        }
                    }
                }
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    if (!((dm >> d) & 1) || !ov) continue;
                    const float wl[12] = {wkp[d][0][0], wkp[d][0][1], wkp[d][0][2], wkp[d][0][3], wkp[d][1][0], wkp[d][1][1], wkp[d][1][2], wkp[d][1][3],
                                          wkp[d][2][0], wkp[d][2][1], wkp[d][2][2], wkp[d][2][3]};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (co0 + i < 7) {
                            const float c = fmaxf(acc3[d][i] + pvp[d][i], 0.f);
                            s0 += wl[i * 3 + 0] * c; s1 += wl[i * 3 + 1] * c; s2 += wl[i * 3 + 2] * c;
                        }
                    }
                }
            } else {
            if constexpr (RB) {                                       // matrix mode 1, net variant: input-slice-major, the fragments of a cell row built once (see the first stage)
#pragma unroll
                for (int d = 0; d < 3; d++) acc3[d] = xt_f4{bias[0], bias[1], bias[2], bias[3]};
                auto body = [&](auto dm_c) {                              // (the item's slices as a compile-time constant: see the first stage)
                    constexpr int DM = decltype(dm_c)::value;
#pragma unroll
                    for (int dd = 0; dd < 3; dd++) {
#pragma unroll
                        for (int kh = 0; kh < 3; kh++) {
                            __builtin_amdgcn_sched_barrier(0);
                            xt_u4 bh, bl;
                            xt_bfrag(xt_smem + am[kh] + dd * ds[kh], bh, bl);
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                const int kd = dd - d + 1;
                                if (kd < 0 || kd > 2 || !((DM >> d) & 1)) continue;
                                acc3[d] = xt_mm3b(wAh[kd * 3 + kh], wAl[kd * 3 + kh], bh, bl, acc3[d]);
                            }
                        }
                    }
                };
                if (dm == 7) body(xt_ic<7>{}); else if (dm == 5) body(xt_ic<5>{}); else body(xt_ic<2>{});
            }
#pragma unroll 1
            for (int d = 0; d < 3; d++) {
                if (!((dm >> d) & 1)) {                               // another wave's slice (wave-uniform)
#pragma unroll
                    for (int kh = 0; kh < 3; kh++) am[kh] += ds[kh];
                    continue;
                }
                // what the epilogue needs, fetched before the instructions run: the residual of the lane's four cells and their 12 align2 weights
                const xt_f4* __restrict__ wk = reinterpret_cast<const xt_f4*>(wt + XT_AL2 + ((d * 64 + h) * 8 + co0) * 3);
                const xt_f4 wk0 = wk[0], wk1 = wk[1], wk2 = wk[2];          // [channel co0 .. co0 + 3][kw]
                float pv[4];
                if constexpr (MM != 0) {
                    xt_load_half((inhw && d < od) ? bufB + ((d * oh + h) * ow + w) * 7 : xt_smem + zidx, co0 >> 2, pv);
                } else {
                    const float* __restrict__ pb = (inhw && d < od) ? bufB + ((d * oh + h) * ow + w) * 7 + co0 : xt_smem + zidx;
#pragma unroll
                    for (int i = 0; i < 4; i++) pv[i] = co0 + i < 7 ? pb[i] : 0.f;
                }
                xt_f4 acc = {bias[0], bias[1], bias[2], bias[3]};
                if constexpr (RB) {
                    acc = d == 0 ? acc3[0] : d == 1 ? acc3[RB ? 1 : 0] : acc3[RB ? 2 : 0];
                } else {
#pragma unroll
                for (int kd = 0; kd < 3; kd++) {
                    const int dd = d + kd - 1;
                    if ((unsigned)dd >= 3u) continue;                 // wave-uniform
#pragma unroll
                    for (int kh = 0; kh < 3; kh++) {
                        const int a = kd == 0 ? am[kh] - ds[kh] : kd == 2 ? am[kh] + ds[kh] : am[kh];
                        if constexpr (MM != 0) {
                            acc = xt_mm3(wAh[kd * 3 + kh], wAl[kd * 3 + kh], xt_smem + a, acc);
                        } else {
#pragma unroll
                            for (int ci = 0; ci < 7; ci++)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[(kd * 3 + kh) * 7 + ci], xt_smem[a + ci], acc, 0, 0, 0);
                        }
                    }
                }
                }
                if (ov) {
                    const float wl[12] = {wk0[0], wk0[1], wk0[2], wk0[3], wk1[0], wk1[1], wk1[2], wk1[3], wk2[0], wk2[1], wk2[2], wk2[3]};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (co0 + i < 7) {
                            const float c = fmaxf(acc[i] + pv[i], 0.f);
                            s0 += wl[i * 3 + 0] * c; s1 += wl[i * 3 + 1] * c; s2 += wl[i * 3 + 2] * c;
                        }
                    }
                }
#pragma unroll
                for (int kh = 0; kh < 3; kh++) am[kh] += ds[kh];
            }
            }       // (!PRE)
            s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);     // channels 0-3 + channels 4-6 of the same cells
            // the tile's pairs lie in rows h0 .. h1: one row at a time, so that the lanes of a pass own distinct columns
            const int h0 = (t * 16) / ppr, h1 = min(t * 16 + 15, npair - 1) / ppr;
            // (rows are serialised by the wave's loop iterations: lanes of different rows update the same LDS word in different iterations.
            // The fence + wave barrier per iteration makes that order a fact of the program, not of the compiler's mood — without it the loop
            // body, which one thread runs at most once, could legally be collapsed to a range test and the rows would collide.)
            for (int r = h0; r <= h1; r++) {
                if (ov && !(q & 1) && h == r) {
                    redw[w * 3 + 0] += s0; redw[w * 3 + 1] += s1; redw[w * 3 + 2] += s2;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
    XT_LAP(6);
    if (tid < 64) {
        const int w = tid;
        float v = kv;
        // out[w] takes column w's middle sums, column w - 1's "right" sums and column w + 1's "left" sums; columns that do not exist are the zero
        // columns 0 and cols + 1 of a wave's array: 48 unconditional reads in flight, added in a fixed order
        const int i1 = min(w, cols) + 1, i0 = min(w, cols + 1), i2 = min(w + 2, cols + 1);
        for (int k = 0; k < nw; k++) {
            const float* rk = red + k * (cols + 2) * 3;
            v += rk[i1 * 3 + 1];
            v += rk[i0 * 3 + 0];
            v += rk[i2 * 3 + 2];
        }
        if (normalize) {          // the reference's row-wise min-max normalisation (baseline/baseline_utils.py:45-63), wave 0 holds the 64 values
            float lo = v, hi = v;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
            float scale = hi - lo;
            if (scale < 1e-5f) scale += 1e-5f;
            v = (v - lo) / scale;
        }
        out[(int64_t)e * 64 + w] = v;
    }
#ifdef XT_PHASE_TIMING
    XT_LAP(7);
    __syncthreads();                                   // (after wave 0 has written the row)
    if (lane == 0) out[(int64_t)e * 64 + 16 + wv] = (float)(xt_w1 - xt_w0);
    if (lane == 0 && !NET) { out[(int64_t)e * 64 + 32 + wv] = (float)(xt_a1 - xt_a0); out[(int64_t)e * 64 + 48 + wv] = (float)(xt_a2 - xt_a1); }
    if (tid == 0) {
        for (int k = 0; k < 7; k++) out[(int64_t)e * 64 + k] = (float)(xt_lap[k + 1] - xt_lap[k]);
        out[(int64_t)e * 64 + 7] = (float)(xt_lap[8] - xt_lap[4]);          // conv 1 stage: the fill loop
        out[(int64_t)e * 64 + 8] = (float)(xt_lap[9] - xt_lap[8]);          //               wave 0's own tiles
        out[(int64_t)e * 64 + 9] = (float)(xt_lap[10] - xt_lap[9]);         //               the next convolution's operands arrive
        out[(int64_t)e * 64 + 10] = (float)(xt_lap[5] - xt_lap[10]);        //               waiting for the other waves
        if (NET) { out[(int64_t)e * 64 + 11] = (float)(xt_lap[11] - xt_lap[3]); out[(int64_t)e * 64 + 12] = (float)(xt_lap[4] - xt_lap[11]); }      // background load / gather
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Actor head (reference baseline/DQN/DQN.py:27-46 `Actor`: mlp 128 -> 128 -> 64 -> 1 with ELU on [state vector ++ net vector]) for every
// legal net of every env, and the greedy action (first maximum in the order of the net-order channel).  One workgroup of 128 threads per
// env; the state half of the first layer is computed once per env; the net vectors come from the per-(region, net) cache
// (agents.NetVectorCache, which must hold every net asked for).  Weights transposed by the caller so that lane j reads element j.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int XA_W1T = 0;                   // [128 in][128 out]  (inputs 0..63: state, 64..127: net vector)
constexpr int XA_B1 = XA_W1T + 128 * 128;
constexpr int XA_W2T = XA_B1 + 128;         // [128 in][64 out]
constexpr int XA_B2 = XA_W2T + 128 * 64;
constexpr int XA_W3 = XA_B2 + 64;           // [64]
constexpr int XA_B3 = XA_W3 + 64;
constexpr int XA_TOTAL = XA_B3 + 1;
constexpr int XA_IDS_MAX = 256;             // net ids of an env staged in LDS (kcap beyond that is refused)

__device__ __forceinline__ float xa_elu(float x) { return x > 0.f ? x : expm1f(x); }

template <bool PRE>       // PRE: the net half of the first layer comes applied (cache_pre): no 32 KB copy of it in LDS — four workgroups per CU instead of two
__global__ void __launch_bounds__(128) xr_actor_kernel(const float* __restrict__ state, const float* __restrict__ head, int64_t head_stride, int ids_off,
                                                       const int32_t* __restrict__ nlegal, const int32_t* __restrict__ region,
                                                       const float* __restrict__ cache_vec, const float* __restrict__ cache_pre, int cache_kmax, const float* __restrict__ wt, int kcap,
                                                       float* __restrict__ logits, int32_t* __restrict__ action,
                                                       const int64_t* __restrict__ env_ids, uint64_t s0) {
    __shared__ float s_w1n[PRE ? 1 : 64 * 128];       // net half of the first layer, [in][out]
    __shared__ float s_w2[128 * 64];
    __shared__ float s_st[64], s_vec[64], s_h1[128];
    const int j = threadIdx.x, e = blockIdx.x;
    if (!PRE) for (int i = j; i < 64 * 128; i += 128) s_w1n[i] = wt[XA_W1T + 64 * 128 + i];
    for (int i = j; i < 128 * 64; i += 128) s_w2[i] = wt[XA_W2T + i];
    if (j < 64) s_st[j] = state[(int64_t)e * 64 + j];
    __syncthreads();
    float hs = wt[XA_B1 + j];
#pragma unroll 8
    for (int i = 0; i < 64; i++) hs += wt[XA_W1T + i * 128 + j] * s_st[i];
    const int nl = min(nlegal[e], kcap);
    const int64_t rbase = (int64_t)region[e] * cache_kmax;
    const float* __restrict__ ids = head + (int64_t)e * head_stride + ids_off;
    const float b2 = j < 64 ? wt[XA_B2 + j] : 0.f, w3 = j < 64 ? wt[XA_W3 + j] : 0.f, b3 = wt[XA_B3];
    float best = -INFINITY;
    int besta = 0;
    // PPO's rollout action (env_ids != null): a sample of Categorical(softmax(logits)) by the Gumbel-max trick, with the counter-based uniform
    // of xroute_env_amd.agents.counter_uniform — the same integer hash of (s0, GLOBAL env id, rank k of the net), the same float operations in
    // the same order (u -> -log(-log(u)) -> logit + g -> first maximum), so the kernel picks what the framework's tensor ops pick
    const uint64_t gid = env_ids ? (uint64_t)env_ids[e] * 16384ull : 0ull;
    auto score = [&](float lg, int k) -> float {
        if (!env_ids) return lg;
        uint64_t x = (gid + (uint64_t)k) ^ s0;
        x = (x ^ ((x >> 30) & 0x3FFFFFFFFull)) * 0x3F58476D1CE4E5B9ull;
        x = (x ^ ((x >> 27) & 0x1FFFFFFFFFull)) * 0x14D049BB133111EBull;
        x = x ^ ((x >> 31) & 0x1FFFFFFFFull);
        const float u = ((float)(uint32_t)((x >> 20) & 0x7FFFFFull) + 0.5f) * (1.0f / 8388608.0f);      // 23 bits: k + 0.5 is exact, u strictly inside (0, 1) (agents.CounterUniform)
        return lg - logf(-logf(u));
    };
    if (PRE) {
        // two nets per round: both waves add their first-layer halves for both nets, then wave 0 runs the second layer of net k and wave 1 that of net
        // k + 1 (every lane busy, half the barriers); thread 0 takes the two logits in order.  The net ids are staged in LDS once and the first-layer
        // products of the NEXT round are fetched while this round computes (as two dependent global loads per round they were most of the kernel).
        __shared__ float s_h1b[2][128], s_lg[2];
        __shared__ int s_ids[XA_IDS_MAX];
        const int wv = j >> 6, o = j & 63;
        const float b2o = wt[XA_B2 + o], w3o = wt[XA_W3 + o];
        for (int k = j; k < nl; k += 128) s_ids[k] = (int)ids[k];
        __syncthreads();
        float c0 = 0.f, c1 = 0.f;
        if (nl > 0) {
            c0 = cache_pre[(rbase + s_ids[0] - 1) * 128 + j];
            c1 = cache_pre[(rbase + s_ids[nl > 1 ? 1 : 0] - 1) * 128 + j];
        }
        for (int k = 0; k < nl; k += 2) {
            const int id0 = s_ids[k], id1 = k + 1 < nl ? s_ids[k + 1] : id0;
            float n0 = 0.f, n1 = 0.f;
            if (k + 2 < nl) {
                n0 = cache_pre[(rbase + s_ids[k + 2] - 1) * 128 + j];
                n1 = cache_pre[(rbase + s_ids[k + 3 < nl ? k + 3 : k + 2] - 1) * 128 + j];
            }
            s_h1b[0][j] = xa_elu(hs + c0);
            s_h1b[1][j] = xa_elu(hs + c1);
            __syncthreads();
            float h2 = b2o;
#pragma unroll 8
            for (int i = 0; i < 128; i++) h2 += s_w2[i * 64 + o] * s_h1b[wv][i];
            float p = w3o * xa_elu(h2);
#pragma unroll
            for (int q = 32; q >= 1; q >>= 1) p += __shfl_xor(p, q, 64);
            if (o == 0) s_lg[wv] = p + b3;
            __syncthreads();
            if (j == 0) {
                const float l0 = s_lg[0], l1 = s_lg[1];
                if (logits) { logits[(int64_t)e * kcap + k] = l0; if (k + 1 < nl) logits[(int64_t)e * kcap + k + 1] = l1; }
                const float s0_ = score(l0, k), s1_ = score(l1, k + 1);
                if (s0_ > best) { best = s0_; besta = id0; }
                if (k + 1 < nl && s1_ > best) { best = s1_; besta = id1; }
            }
            c0 = n0; c1 = n1;
        }
    } else {
    for (int k = 0; k < nl; k++) {
        const int id = (int)ids[k];
        float h1 = hs;
        if (j < 64) s_vec[j] = cache_vec[(rbase + id - 1) * 64 + j];
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 64; i++) h1 += s_w1n[i * 128 + j] * s_vec[i];
        s_h1[j] = xa_elu(h1);
        __syncthreads();
        if (j < 64) {
            float h2 = b2;
#pragma unroll 8
            for (int i = 0; i < 128; i++) h2 += s_w2[i * 64 + j] * s_h1[i];
            float p = w3 * xa_elu(h2);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
            const float lg = p + b3;
            if (j == 0) {
                if (logits) logits[(int64_t)e * kcap + k] = lg;
                const float sc = score(lg, k);
                if (sc > best) { best = sc; besta = id; }
            }
        }
    }
    }
    if (logits) for (int k = nl + j; k < kcap; k += 128) logits[(int64_t)e * kcap + k] = -INFINITY;
    if (j == 0) action[e] = besta;
}

// matrix mode of the towers' 7 -> 7-channel convolutions: 1 (default) = split-bf16 operands on v_mfma_f32_16x16x32_bf16 (three instructions per product, fp32
// accumulation: see xt_mm3), 0 = fp32 operands on v_mfma_f32_16x16x4_f32 (XR_TOWER_FP32=1: the exact-fp32 form of rounds 4-5, kept for A/B and as the reference
// of the split form's tolerance tests)
static int xt_matrix_mode() {
    static const int mm = [] { const char* v = getenv("XR_TOWER_FP32"); return (v && v[0] == '1') ? 0 : 1; }();
    return mm;
}

}  // namespace

extern "C" {

int32_t xr_agent_matrix_mode(void) { return xt_matrix_mode(); }

int32_t xr_agent_obstacle_tower_weights(void) { return XT_TOTAL; }

// head_dev: fp32 [n_envs][head_stride], plane 0 of every env's observation first (xr_batch_step_compact / _observe buffers);
// dims = the observation's (D, H, W) as the reference's networks see it (the tensor is [.., Z, Y, X]: D = dim_z, H = dim_y, W = dim_x);
// weights_dev: xr_agent_obstacle_tower_weights() floats, packed by xroute_env_amd/agents.py FusedObstacleTower; out_dev: fp32 [n_envs][64].
// Returns XR_ERR_RANGE for grids this kernel does not take (the caller keeps the library path for those).
int32_t xr_agent_obstacle_tower(const float* head_dev, int64_t head_stride, int32_t n_envs, int32_t D, int32_t H, int32_t W,
                                const float* weights_dev, float* out_dev, int32_t normalize, void* stream) {
    if (!head_dev || !weights_dev || !out_dev || n_envs < 0 || D < 1 || H < 1 || W < 1) return XR_ERR_INVALID;
    if (n_envs == 0) return XR_OK;
    XtDims g;
    g.D = D; g.H = H; g.W = W;
    auto strd = [](int s, int t) { return (s > t ? (s - t + t - 1) / t : 0) + 1; };           // ceil(max(0, s - t) / t) + 1
    g.sd = strd(D, 3); g.sh = strd(H, 64); g.sw = strd(W, 64);
    if (D + 2 < 5 || H + 2 < 5 || W + 2 < 5) return XR_ERR_RANGE;
    g.od = (D + 2 - 5) / g.sd + 1; g.oh = (H + 2 - 5) / g.sh + 1; g.ow = (W + 2 - 5) / g.sw + 1;
    if (g.od > 3 || g.oh + 3 > 64 || g.ow + 3 > 64) return XR_ERR_RANGE;
    g.cols = g.ow + 2;
    const int64_t N = (int64_t)D * H * W, nB = 7LL * g.od * g.oh * g.ow, nC1 = 21LL * (g.oh + 3) * (g.ow + 3);
    // threads per workgroup (one workgroup per CU either way: LDS): matrix mode 1 runs 768 — three waves per SIMD have 170 registers each, what the second 7 -> 7
    // stage needs to build every fragment once (three accumulators + the three slices' epilogue operands asked for up front): 0.155 against 0.163 ms per 1024 envs;
    // the fp32 mode keeps 1024 (0.83 against 0.93 ms per agent step, round 5).  XR_TOWER_THREADS overrides (A/B).
    static const int threads = [] { const char* v = getenv("XR_TOWER_THREADS"); const int t = v ? atoi(v) : (xt_matrix_mode() ? 768 : 1024); return (t == 256 || t == 512 || t == 768 || t == 1024) ? t : 1024; }();
    const int64_t Np = (int64_t)(D + 2) * (H + 2) * ((W + 2) | 1);
    g.y_in_b = Np <= nB;
    const int64_t c1_alloc = nC1 > (g.y_in_b ? Np : 2 * Np) ? nC1 : (g.y_in_b ? Np : 2 * Np);      // floats behind b: the first activation of the 7-channel block, or
                                                                                               // the 1-channel block's grids where they need more (narrow regions)
    {   // cells per thread of the 1-channel convolutions: 9 (S + 2) reads + 27 S FMAs + ~25 index instructions per strip, times the passes; 4 = the packed form
        // (xt_conv1_pk4: 9 x 11 instructions + the same overhead, measured 1.14 x the cost of a strip of three per pass)
        auto passes = [&](int S) { const int64_t strips = (int64_t)D * H * ((W + S - 1) / S); return (strips + threads - 1) / threads; };
        const int64_t c3 = passes(3) * (36 * 3 + 43), c4 = passes(4) * 172, c5 = passes(5) * (36 * 5 + 43);
        g.strip = c4 <= c3 && c4 <= c5 ? 4 : c5 < c3 ? 5 : 3;
        if (const char* v = getenv("XR_TOWER_STRIP")) { const int S = atoi(v); if (S == 3 || S == 4 || S == 5) g.strip = S; }      // (A/B)
    }
    g.vec_load = (W % 4 == 0) && (head_stride % 4 == 0) && (reinterpret_cast<uintptr_t>(head_dev) % 16 == 0);
    if (nB < 1024 * 3 || head_stride < N) return XR_ERR_RANGE;
    g.tail = (int)(nB + c1_alloc);
    const size_t lds = (size_t)(nB + c1_alloc + 8 + (threads / 64) * (g.cols + 2) * 3) * sizeof(float);
    if (lds > 160 * 1024) return XR_ERR_RANGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const XnArgs none{};
    const int mm = xt_matrix_mode();
    hipError_t e = hipSuccess;
#define XT_LAUNCH(BT, MM)                                                                                                                                   \
    do {                                                                                                                                                    \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xr_ob_tower_kernel<BT, false, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        if (e == hipSuccess) {                                                                                                                              \
            hipLaunchKernelGGL((xr_ob_tower_kernel<BT, false, MM>), dim3(n_envs), dim3(BT), lds, st, head_dev, head_stride, n_envs, g, weights_dev, out_dev, normalize, none); \
            e = hipGetLastError();                                                                                                                          \
        }                                                                                                                                                   \
    } while (0)
    if (threads == 1024) { if (mm) XT_LAUNCH(1024, 1); else XT_LAUNCH(1024, 0); }
    else if (threads == 768) { if (mm) XT_LAUNCH(768, 1); else XT_LAUNCH(768, 0); }
    else if (threads == 512) { if (mm) XT_LAUNCH(512, 1); else XT_LAUNCH(512, 0); }
    else { if (mm) XT_LAUNCH(256, 1); else XT_LAUNCH(256, 0); }
#undef XT_LAUNCH
    return e == hipSuccess ? XR_OK : XR_ERR_HIP;
}

int32_t xr_agent_net_tower_weights(void) { return XN_TOTAL; }

// The net tower of the reference's RepresentationNetwork (`net_conv1 -> net_align_conv1 -> net_conv2 -> net_align_conv2`, baseline/baseline_utils.py:246-262,
// 350-357) for n_pairs (region, net) pairs of ONE grid shape, straight from the regions' access-point lists: the same kernel as the obstacle tower with a sparse
// front end (see XnArgs / the NET branch).  Called by xr_batch_net_vectors (xr_batch.cpp), which owns the region tables.  bg_dev: fp32 [od*oh*ow*7];
// out_dev fp32 [n_pairs][64]; flags_dev int32 [n_pairs] (zeroed by the caller): 1 = the net's lists do not fit LDS, 2 = wrong shape / no such net — those rows
// of out_dev are left unwritten.
hipError_t xr_launch_net_tower(const void* regions, const int32_t* net_csr, const int32_t* ap_feat, int32_t n_regions, const int32_t* pair_region_dev,
                               const int32_t* pair_net_dev, int32_t n_pairs, int32_t D, int32_t H, int32_t W, const float* weights_dev, const float* bg_dev,
                               float* out_dev, int32_t* flags_dev, int32_t normalize, hipStream_t st, int32_t* status) {
    *status = XR_OK;
    if (n_pairs == 0) return hipSuccess;
    XtDims g;
    g.D = D; g.H = H; g.W = W;
    auto strd = [](int s, int t) { return (s > t ? (s - t + t - 1) / t : 0) + 1; };
    g.sd = strd(D, 3); g.sh = strd(H, 64); g.sw = strd(W, 64);
    if (D + 2 < 5 || H + 2 < 5 || W + 2 < 5) { *status = XR_ERR_RANGE; return hipSuccess; }
    g.od = (D + 2 - 5) / g.sd + 1; g.oh = (H + 2 - 5) / g.sh + 1; g.ow = (W + 2 - 5) / g.sw + 1;
    if (g.od > 3 || g.oh + 3 > 64 || g.ow + 3 > 64 || (int64_t)D * H * W >= 65536) { *status = XR_ERR_RANGE; return hipSuccess; }
    g.cols = g.ow + 2;
    const int64_t nB = 7LL * g.od * g.oh * g.ow, nC1 = 21LL * (g.oh + 3) * (g.ow + 3);
    const int64_t fixed = 6 * ((int64_t)D * H) + 256 + 8 + 112 + 952;      // (the NET front end's fixed LDS part: one mask word per (d, h) row)
    if (nB < 1024 * 3 || nC1 < fixed + 1024 || W > 32 || D > 32 || H > 64 || g.sh != 1 || g.sw != 1) { *status = XR_ERR_RANGE; return hipSuccess; }      // (packed coordinates 5 + 6 + 5 bits)
    g.y_in_b = 1; g.strip = 3; g.vec_load = 0;
    g.tail = (int)(nB + nC1);
    const size_t lds = (size_t)(nB + nC1 + 8 + 16 * (g.cols + 2) * 3) * sizeof(float);
    if (lds > 160 * 1024) { *status = XR_ERR_RANGE; return hipSuccess; }
    const int mm = xt_matrix_mode();
    const void* fn = mm ? reinterpret_cast<const void*>(&xr_ob_tower_kernel<1024, true, 1>) : reinterpret_cast<const void*>(&xr_ob_tower_kernel<1024, true, 0>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    XnArgs na;
    na.regions = static_cast<const XrRegionDev*>(regions); na.net_csr = net_csr; na.ap_feat = ap_feat; na.pair_region = pair_region_dev; na.pair_net = pair_net_dev;
    na.bg = bg_dev; na.flags = flags_dev; na.n_regions = n_regions;
    if (mm) hipLaunchKernelGGL((xr_ob_tower_kernel<1024, true, 1>), dim3(n_pairs), dim3(1024), lds, st, static_cast<const float*>(nullptr), (int64_t)0, n_pairs, g, weights_dev, out_dev, normalize, na);
    else hipLaunchKernelGGL((xr_ob_tower_kernel<1024, true, 0>), dim3(n_pairs), dim3(1024), lds, st, static_cast<const float*>(nullptr), (int64_t)0, n_pairs, g, weights_dev, out_dev, normalize, na);
    return hipGetLastError();
}

int32_t xr_agent_actor_weights(void) { return XA_TOTAL; }

// state_dev fp32 [n_envs][64] (normalised state vectors), head_dev as above (the net-order channel = plane 1 starts at float ids_off of an env's row),
// nlegal_dev / region_dev int32 [n_envs], cache_vec_dev fp32 [regions * cache_kmax][64] (normalised net vectors, row = region * cache_kmax + net - 1),
// cache_pre_dev (optional) fp32 [regions * cache_kmax][128]: the net half of the first layer already applied to every cached vector (W1[:, 64:] . vec),
// weights_dev: xr_agent_actor_weights() floats; logits_dev (optional) fp32 [n_envs][kcap] (-inf beyond an env's nets); action_dev int32 [n_envs]:
// the greedy net id (0: no nets).
int32_t xr_agent_actor(const float* state_dev, const float* head_dev, int64_t head_stride, int32_t ids_off, const int32_t* nlegal_dev,
                       const int32_t* region_dev, const float* cache_vec_dev, const float* cache_pre_dev, int32_t cache_kmax, const float* weights_dev,
                       int32_t n_envs, int32_t kcap, float* logits_dev, int32_t* action_dev, void* stream) {
    return xr_agent_actor_sample(state_dev, head_dev, head_stride, ids_off, nlegal_dev, region_dev, cache_vec_dev, cache_pre_dev, cache_kmax, weights_dev,
                                 n_envs, kcap, logits_dev, action_dev, nullptr, 0ull, stream);
}

// The same with PPO's rollout sampling inside (baseline/PPO/PPO.py:124-146, 205-217: `dist = Categorical(action_probs); action = dist.sample()`):
// env_ids_dev int64 [n_envs] = the GLOBAL id of every env, s0 = the (seed, step) mix of xroute_env_amd.agents.counter_uniform -> action_dev = a sample of
// Categorical(softmax(logits)) by the Gumbel-max trick with counter-based uniforms (no generator state: the same env gets the same action whichever
// rank or batch evaluates it).  env_ids_dev == NULL: the greedy action (xr_agent_actor).
int32_t xr_agent_actor_sample(const float* state_dev, const float* head_dev, int64_t head_stride, int32_t ids_off, const int32_t* nlegal_dev,
                              const int32_t* region_dev, const float* cache_vec_dev, const float* cache_pre_dev, int32_t cache_kmax, const float* weights_dev,
                              int32_t n_envs, int32_t kcap, float* logits_dev, int32_t* action_dev, const int64_t* env_ids_dev, uint64_t s0, void* stream) {
    if (!state_dev || !head_dev || !nlegal_dev || !region_dev || !cache_vec_dev || !weights_dev || !action_dev || n_envs < 0 || kcap < 1 || cache_kmax < 1 ||
        ids_off < 0 || head_stride < (int64_t)ids_off + kcap)
        return XR_ERR_INVALID;
    if (cache_pre_dev && kcap > XA_IDS_MAX) return XR_ERR_RANGE;
    if (n_envs == 0) return XR_OK;
    if (cache_pre_dev)
        hipLaunchKernelGGL(xr_actor_kernel<true>, dim3(n_envs), dim3(128), 0, static_cast<hipStream_t>(stream), state_dev, head_dev, head_stride, ids_off, nlegal_dev,
                           region_dev, cache_vec_dev, cache_pre_dev, cache_kmax, weights_dev, kcap, logits_dev, action_dev, env_ids_dev, s0);
    else
        hipLaunchKernelGGL(xr_actor_kernel<false>, dim3(n_envs), dim3(128), 0, static_cast<hipStream_t>(stream), state_dev, head_dev, head_stride, ids_off, nlegal_dev,
                           region_dev, cache_vec_dev, cache_pre_dev, cache_kmax, weights_dev, kcap, logits_dev, action_dev, env_ids_dev, s0);
    return hipGetLastError() == hipSuccess ? XR_OK : XR_ERR_HIP;
}

}  // extern "C"
