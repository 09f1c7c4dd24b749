// field_bwd.hip -- gradients of the per-sample field (BASELINE.json configs[4]: training step through the HIP ops).
//
// Layer-wise backward over the activations saved by field_forward_f32_kernel<SAVE> (sahs::act layout: one dense
// [P x width] array per layer): for every dense layer   dW += dY^T X (+ db = column sums of dY, fused),   dX = (dY W) * act'(X)
// run as fp32 MFMA GEMMs over all samples of the call (gemm_dma_kernel: operand tiles by LDS-DMA, epilogues staged through
// LDS), plus small kernels for the positional-encoding, tanh and trilinear-grid derivatives.  Built once per model
// (sahs_model.hpp).  Not fused across layers: activations and per-layer gradients go through HBM.
// Conventions follow autograd of the reference graph
// (models.py:514-528, modules.py:371-390 / 444-462 / 254-295): the per-frame constant inputs (driving, pose
// encoding) get their weight-column gradients from the bias gradient (their value is the same for every sample:
// dW[:, const] = db (x) c) and their own gradient from W[:, const]^T db.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"

namespace SAHS_NS {

// ------------------------------------------------------------------------------------------------
// C[M x N] (op)= A'[M x K] B[K x N], fp32, v_mfma_f32_16x16x4_f32.  TA: A'(m,k) = A[k*lda + m], else A[m*lda + k];
// B(k,n) = B[k*ldb + n].  64x64 tile per 256-thread workgroup, K in steps of 16 through LDS; blockIdx.z splits K.
// mode 0: C = acc; 1: C += acc; 2: atomicAdd(C, acc).  mask != null: acc *= (mask[m*ldm + n] > 0 ? 1 : slope).
// ------------------------------------------------------------------------------------------------
constexpr int GT = 128, GK = 16, GLD = 132;   // 128x128 tile, K-step 16; LDS row stride 132 floats

// Global -> register staging of one K-step of both operands (8 floats per thread per operand), so that the loads of step
// k+1 are in flight while step k is multiplied (the LDS tiles are double buffered).
struct Stage { float a[8], b[8]; };

template <bool TA>
__device__ __forceinline__ void stage_load(Stage &st, int tid, long m0, int n0, long k0, long k_hi, int M, int N, const float *__restrict__ A,
                                           long lda, const float *__restrict__ B, long ldb, bool va, bool vb)
{
    // A' tile element (kk, mm): thread covers 8 consecutive elements along the contiguous global direction
    if (TA) {   // A'(m,k) = A[k*lda + m]: contiguous in m.  16 k-rows x 128 m: thread -> row kk = tid/16, mm = (tid%16)*8
        const int kk = tid >> 4, mm = (tid & 15) * 8;
        const long k = k0 + kk, m = m0 + mm;
        if (va && k < k_hi && m + 7 < M) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(A + k * lda + m), v1 = *reinterpret_cast<const f32x4 *>(A + k * lda + m + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { st.a[i] = v0[i]; st.a[4 + i] = v1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st.a[i] = (k < k_hi && m + i < M) ? A[k * lda + m + i] : 0.0f;
        }
    } else {    // A'(m,k) = A[m*lda + k]: contiguous in k.  128 m-rows x 16 k: thread -> mm = tid/2, kk = (tid%2)*8
        const int mm = tid >> 1, kk = (tid & 1) * 8;
        const long m = m0 + mm, k = k0 + kk;
        if (va && m < M && k + 7 < k_hi) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(A + m * lda + k), v1 = *reinterpret_cast<const f32x4 *>(A + m * lda + k + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { st.a[i] = v0[i]; st.a[4 + i] = v1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st.a[i] = (m < M && k + i < k_hi) ? A[m * lda + k + i] : 0.0f;
        }
    }
    {           // B(k,n) = B[k*ldb + n]: contiguous in n.  16 k-rows x 128 n
        const int kk = tid >> 4, nn = (tid & 15) * 8;
        const long k = k0 + kk;
        const int n = n0 + nn;
        if (vb && k < k_hi && n + 7 < N) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(B + k * ldb + n), v1 = *reinterpret_cast<const f32x4 *>(B + k * ldb + n + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { st.b[i] = v0[i]; st.b[4 + i] = v1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st.b[i] = (k < k_hi && n + i < N) ? B[k * ldb + n + i] : 0.0f;
        }
    }
}

template <bool TA>
__device__ __forceinline__ void stage_store(const Stage &st, int tid, float (*As)[GLD], float (*Bs)[GLD])
{
    if (TA) {
        const int kk = tid >> 4, mm = (tid & 15) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) As[kk][mm + i] = st.a[i];
    } else {
        const int mm = tid >> 1, kk = (tid & 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) As[kk + i][mm] = st.a[i];
    }
    const int kk = tid >> 4, nn = (tid & 15) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) Bs[kk][nn + i] = st.b[i];
}

// TA && rowsum != null: the n-block-0 workgroups also accumulate rowsum[m] += sum_k A'(m,k) (the bias gradient of the layer whose
// weight gradient this GEMM is) from the A tiles they stage anyway.
// nbn > 0 (1-D grid): XCD-aware mapping -- the nbn column blocks of one row block run on the same XCD (consecutive slots of
// workgroup ids congruent mod 8), so the second one finds the shared A tile in that XCD's L2.
template <bool TA>
__global__ void __launch_bounds__(256) gemm_f32_kernel(int M, int N, int K, const float *__restrict__ A, long lda,
                                                       const float *__restrict__ B, long ldb, float *__restrict__ C, long ldc, int mode,
                                                       const float *__restrict__ mask, long ldm, float slope, int kslab,
                                                       float *__restrict__ rowsum, int nbn)
{
    __shared__ float As[2][GK][GLD], Bs[2][GK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;              // each wave: 64 x 64 = 4 x 4 MFMA tiles
    int bx = blockIdx.x, by = blockIdx.y;
    if (nbn > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bx = slot % nbn;
        by = (slot / nbn) * 8 + xcd;
        if ((long)by * GT >= M) return;
    }
    const long m0 = (long)by * GT;
    const int n0 = bx * GT;
    const long k_lo = (long)blockIdx.z * kslab;
    const bool do_sum = TA && rowsum != nullptr && bx == 0;
    float cs[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const long k_hi = (k_lo + kslab < K) ? k_lo + kslab : K;
    // 16-byte vector loads only when every row start is 16-byte aligned
    const bool va = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
    const bool vb = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    Stage st;
    stage_load<TA>(st, tid, m0, n0, k_lo, k_hi, M, N, A, lda, B, ldb, va, vb);
    stage_store<TA>(st, tid, As[0], Bs[0]);
    if (do_sum)
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[i] += st.a[i];
    __syncthreads();
    int cur = 0;
    for (long k0 = k_lo; k0 < k_hi; k0 += GK) {
        const bool more = k0 + GK < k_hi;
        if (more) {
            stage_load<TA>(st, tid, m0, n0, k0 + GK, k_hi, M, N, A, lda, B, ldb, va, vb);
            if (do_sum)
#pragma unroll
                for (int i = 0; i < 8; ++i) cs[i] += st.a[i];
        }
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[cur][4 * s + q][64 * wm + 16 * i + c16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[cur][4 * s + q][64 * wn + 16 * j + c16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage_store<TA>(st, tid, As[cur ^ 1], Bs[cur ^ 1]);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + 64 * wm + 16 * i + 4 * q + r;
                const int n = n0 + 64 * wn + 16 * j + c16;
                if (m < M && n < N) {
                    float v = acc[i][j][r];
                    if (mask != nullptr) v *= (mask[m * ldm + n] > 0.0f) ? 1.0f : slope;
                    float *dst = C + m * ldc + n;
                    if (mode == 0) *dst = v;
                    else if (mode == 1) *dst += v;
                    else atomicAdd(dst, v);
                }
            }
    if (do_sum) {   // the K loop ended on a barrier: the LDS tiles are free.  Thread (kk, mm) holds the sums of its 8 m over k = kk mod 16.
        const int kk = tid >> 4, mm = (tid & 15) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) As[0][kk][mm + i] = cs[i];
        __syncthreads();
        if (tid < GT && m0 + tid < M) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < GK; ++k) t += As[0][k][tid];
            atomicAdd(rowsum + m0 + tid, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same two GEMMs with the operand tiles moved global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR
// staging, no ds_write), three K-steps deep: while step t is multiplied, steps t+1 and t+2 are in flight.  One raw
// s_barrier per K-step; the wait before it is a counted vmcnt (4 = the wave's DMA instructions of the one newer step).
// Needs 16-byte aligned operand rows (A, B, lda, ldb multiples of 4 floats) -- the host routes everything else to
// gemm_f32_kernel above.  Out-of-range rows/columns are fetched from a 16-byte zero page.
//   TA : A'(m,k) = A[k*lda + m]  tile [16 k][128 m]          B(k,n) = B[k*ldb + n]  tile [16 k][128 n]
//   !TA: A'(m,k) = A[m*lda + k]  tile [128 m][16 k] (one ds_read_b128 = 4 k of one row; the MFMA k index is
//        permuted -- lane q takes k = 4q + r at MFMA r -- identically for both operands, so the sum is unchanged)
// K-major tiles are XOR-swizzled by 16 floats on alternate rows (TA: k&1, !TA: (k>>2)&1 -- either way q&1 on the MFMA
// side), so the four q groups of a ds_read_b32 hit two disjoint bank sets.
// ------------------------------------------------------------------------------------------------
// X3: the products on the bf16 matrix pipe at near-fp32 accuracy (the operand split of field_bf16x3.hip): every fp32 operand value is
// split on its way from LDS to the MFMA into x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16-17 significant bits together) and a product
// is three v_mfma_f32_32x32x16_bf16 with fp32 accumulation, a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped a_lo b_lo term is ~2^-18
// of the product).  One 16-sample (TA) / 16-feature (!TA) K-step of the LDS ring is exactly one MFMA k-depth; a wave's 64 x 64 block is
// 2 x 2 tiles of 32 x 32, 12 MFMAs per K-step (384 matrix-pipe cycles against the 2048 of the 64 f32 MFMAs it replaces), so these GEMMs
// turn from matrix-bound into HBM-bound; the ~100 VALU instructions of the splits per K-step ride in the slack.  Same tiles, ring,
// epilogues and arguments as the f32 form (X3 = false), which stays selectable (SAHS_BWD_GEMM=f32) as the exact A/B reference.
typedef __attribute__((address_space(3))) void *lds_void_t;
typedef const __attribute__((address_space(1))) void *gbl_void_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// ring depth: 3 for the data-gradient GEMM (48 KB of LDS, three workgroups per CU, 16-step K loops), 4 for the weight-gradient GEMM (64 KB,
// two workgroups per CU by its registers anyway, K loops of 32..512 steps: three K-steps = 48 KB per workgroup in flight -- the
// kernel is HBM-bound and at depth 3 it reached 3.7 TB/s with 64 KB per CU in flight)
template <bool TA> constexpr int ring_depth() { return TA ? 4 : 3; }
constexpr int DTILE = GK * GT;                // floats per operand tile (8 KB)

// workgroup barrier that orders LDS traffic only: __syncthreads() is a full fence and makes the compiler drain vmcnt too, i.e. wait for every
// outstanding global STORE to be acknowledged -- not needed where the barrier only protects a staging buffer in LDS
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// eight fp32 values (this lane's k = 8h .. 8h+7 of one row/column) -> the bf16x8 MFMA fragments of their hi and lo parts
__device__ __forceinline__ void split8(const float (&x)[8], u32x4_t &hi, u32x4_t &lo)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{x[2 * p], x[2 * p + 1]}, bf16x2_t));
        hi[p] = h;
        lo[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{x[2 * p] - __builtin_bit_cast(float, h << 16),
                                                                             x[2 * p + 1] - __builtin_bit_cast(float, h & 0xffff0000u)}, bf16x2_t));
    }
}
__device__ __forceinline__ f32x16_t mfma3(const u32x4_t &ah, const u32x4_t &al, const u32x4_t &bh, const u32x4_t &bl, f32x16_t c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh), c, 0, 0, 0);
    return c;
}

template <bool TA, bool X3>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((TA && X3) ? 2 : 3, 3))) gemm_dma_kernel(int M, int N, int K, const float *__restrict__ A, long lda,
                                                       const float *__restrict__ B, long ldb, float *__restrict__ C, long ldc, int mode,
                                                       const float *__restrict__ mask, long ldm, float slope, int kslab,
                                                       float *__restrict__ rowsum, int nbn, const float *__restrict__ zero,
                                                       unsigned char *__restrict__ bits)
{
    // bits (optional): the sign pattern of an activation matrix as a bit matrix, row r = N/8 bytes, bit (n & 7) of byte n >> 3 set where
    // X[r][n] > 0.  The weight-gradient GEMM (TA) WRITES it for its B operand X from the tiles it stages anyway (the m-block-0 workgroups);
    // the data-gradient GEMM of the same layer, which follows it and whose (leaky-)ReLU mask is that same X, READS it instead of the fp32
    // matrix -- 1/32 of the bytes of what was a third of its HBM traffic.
    constexpr int DST = ring_depth<TA>();
    __shared__ __attribute__((aligned(16))) float smem[DST][2][DTILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    long bz = blockIdx.z;
    if (nbn > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bx = slot % nbn;
        by = (slot / nbn) * 8 + xcd;
        if ((long)by * GT >= M) return;
    } else if (nbn < 0) {
        // weight-gradient GEMM, 1-D grid: the nx x ny output tiles of ONE sample slab read the same dY and X rows -- they run on the same
        // XCD (workgroup ids congruent mod 8, consecutive slots), so that XCD's L2 fetches the slab from HBM once instead of once per tile
        const int nx = -nbn, ny = (M + GT - 1) / GT, tiles = nx * ny;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int tile = slot % tiles;
        bz = (long)(slot / tiles) * 8 + xcd;
        bx = tile % nx;
        by = tile / nx;
        if (bz * kslab >= K) return;
    }
    const long m0 = (long)by * GT;
    const int n0 = bx * GT;
    const long k_lo = bz * kslab;
    const long k_hi = (k_lo + kslab < K) ? k_lo + kslab : K;
    const int T = (int)((k_hi - k_lo + GK - 1) / GK);
    const bool do_sum = TA && rowsum != nullptr && bx == 0;

    // ---- this wave's four DMA instructions per K-step: waves 0,1 move the A tile, waves 2,3 the B tile ----
    const int h = lane >> 5, c32 = lane & 31;
    const float *src[4]; long step[4]; int krow[4]; bool colok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rp = (wave & 1) * 4 + u;                 // 0..7: which 1-KB eighth of the tile
        if (X3 && !TA && wave >= 2) {                        // B = pre-split, fragment-ordered weight pieces: a linear 8 KB copy per K-step
            src[u] = B + ((k_lo / GK) * (long)nbn + bx) * (GK * GT) + (rp * 64 + lane) * 4;
            step[u] = (long)nbn * (GK * GT);
            krow[u] = 0;
            colok[u] = true;
        } else if (wave < 2 && !TA) {                        // A, row-major [128 m][16 k]: 4 chunks per row
            const int chunk = rp * 64 + lane, m = chunk >> 2, kc = chunk & 3;
            src[u] = A + (m0 + m) * lda + k_lo + 4 * kc;
            step[u] = GK;
            krow[u] = 4 * kc;                                // valid while k0 + krow < k_hi
            colok[u] = m0 + m < M;
        } else {                                             // K-major [16 k][128 cols], 32 chunks per row, two rows per instruction
            const float *base = (wave < 2) ? A : B;
            const long ld = (wave < 2) ? lda : ldb;
            const long c0 = (wave < 2) ? m0 : n0;
            const int dim = (wave < 2) ? M : N;
            const int row = 2 * rp + h;
            const int sw = TA ? (row & 1) : ((row >> 2) & 1);
            const int j = c32 ^ (sw << 2);
            src[u] = base + (k_lo + row) * ld + c0 + 4 * j;
            step[u] = GK * ld;
            krow[u] = row;
            colok[u] = c0 + 4 * j < dim;
        }
    }
    const int dst_off = ((wave < 2) ? 0 : DTILE) + (wave & 1) * 4 * 256;   // + u*256 floats, + lane*4 by the hardware
    auto issue = [&](int t) {
        float *dstb = &smem[t % DST][0][0] + dst_off;
        const long k0 = k_lo + (long)t * GK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *g = (colok[u] && k0 + krow[u] < k_hi) ? src[u] + (long)t * step[u] : zero;
            __builtin_amdgcn_global_load_lds((gbl_void_t)g, (lds_void_t)(dstb + u * 256), 16, 0, 0);
        }
    };

    // ---- MFMA-side LDS offsets ----
    int colA[4], colB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = 64 * wm + 16 * i + c16, n = 64 * wn + 16 * i + c16;
        colA[i] = TA ? ((((m >> 2) ^ ((q & 1) << 2)) << 2) + (m & 3)) : (m * GK + 4 * q);
        colB[i] = (((n >> 2) ^ ((q & 1) << 2)) << 2) + (n & 3);
    }
    // X3: lane (r32, h) of a 32x32x16 MFMA supplies k = 8h .. 8h+7 of row/column r32 of a 32-wide tile.  K-major tiles are swizzled on
    // alternate rows (TA: k & 1, !TA: (k >> 2) & 1), so a lane's eight k sit in two column positions: x3c[t][0] where the swizzle bit is
    // clear, x3c[t][1] where it is set.  (Row-major A of !TA: the eight k are 32 contiguous bytes of row m.)
    const int r32 = lane & 31;
    int x3a[2][2], x3b[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = 64 * wm + 32 * t + r32, n = 64 * wn + 32 * t + r32;
        x3a[t][0] = TA ? (((m >> 2) << 2) + (m & 3)) : (m * GK + 8 * h);
        x3a[t][1] = TA ? ((((m >> 2) ^ 4) << 2) + (m & 3)) : 0;
        x3b[t][0] = ((n >> 2) << 2) + (n & 3);
        x3b[t][1] = (((n >> 2) ^ 4) << 2) + (n & 3);
    }
    f32x4 acc[4][4];
    f32x16_t acx[2][2];
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acx[i][j][e] = 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float cs = 0.0f;

    // vmcnt accounting of the K loop (audited on the ISA, tools/check_isa.py; the build fails if this kernel gets scratch): the only
    // vector-memory instructions a wave executes between kernel entry and the end of the loop are its 4 LDS-DMA instructions per
    // issue(); they complete in issue order, so "vmcnt(4)" at step t means step t has landed while step t+1 may still be in flight --
    // and any further VMEM instruction the compiler might ever add there could only make that wait stricter, never weaker.
    // (depth DST: steps t+1 .. t+DST-2 may be in flight at step t's wait -- 4 DMA instructions each -- and step t+DST-1 is issued after it)
    // data-gradient GEMM: this lane's 16 mask bytes of the epilogue (row (tid >> 5) + 8 e, columns n0 + 4 (tid & 31) ..) are requested
    // here, ahead of the K loop -- they are older than every LDS-DMA, so the counted waits below stay correct (stricter) -- and have long
    // landed when the epilogue packs them; fetched after the loop they cost every workgroup an exposed memory round trip
    unsigned char pre_raw[16];
    const bool pre_bits = !TA && mode != 2 && mask != nullptr && bits != nullptr;
    if constexpr (!TA) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long m = m0 + (tid >> 5) + 8 * e;
            const int en = n0 + 4 * (tid & 31);
            pre_raw[e] = pre_bits ? bits[(m < M && en < N) ? m * (N >> 3) + (en >> 3) : 0] : (unsigned char)0;
        }
    }
    if (T > 0) issue(0);      // an empty K slab (k_lo >= K) issues nothing and falls through to a zero contribution
    if (T > 1) issue(1);
    if (DST > 3 && T > 2) issue(2);
    for (int t = 0; t < T; ++t) {
        if (DST > 3 && t + 2 < T) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + DST - 1 < T) issue(t + DST - 1);          // into the buffer every wave finished reading before the barrier above
        const float *As = &smem[t % DST][0][0], *Bs = &smem[t % DST][1][0];
        if (TA && bits != nullptr && by == 0) {      // thread (kk, g): the 8 columns 8g .. 8g+7 of sample row kk of the X tile -> one byte
            const int kk = tid >> 4, g = tid & 15;
            const long k = k_lo + (long)t * GK + kk;
            if (k < k_hi && n0 + 8 * g < N) {
                const float *px = Bs + kk * GT + 4 * ((2 * g) ^ ((kk & 1) << 2));     // chunks 2g, 2g+1 stay adjacent under the swizzle
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(px), v1 = *reinterpret_cast<const f32x4 *>(px + 4);
                unsigned b8 = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) b8 |= (v0[i] > 0.0f ? 1u : 0u) << i | (v1[i] > 0.0f ? 1u : 0u) << (4 + i);
                bits[k * (N >> 3) + (n0 >> 3) + g] = (unsigned char)b8;
            }
        }
        if constexpr (X3) {
            u32x4_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float x[8];
                if (TA) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = As[(8 * h + j) * GT + x3a[i][j & 1]];
                } else {
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(As + x3a[i][0]), v1 = *reinterpret_cast<const f32x4 *>(As + x3a[i][0] + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
                }
                split8(x, ah[i], al[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (TA) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = Bs[(8 * h + j) * GT + x3b[i][j & 1]];
                    split8(x, bh[i], bl[i]);
                } else {      // weights arrive split and in fragment order (CopyJob pack): n-tile 2 wn + i, hi then lo
                    bh[i] = *reinterpret_cast<const u32x4_t *>(Bs + ((2 * wn + i) * 64 + lane) * 4);
                    bl[i] = *reinterpret_cast<const u32x4_t *>(Bs + GK * GT / 2 + ((2 * wn + i) * 64 + lane) * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acx[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acx[i][j]);
            if (TA && do_sum && tid < GT) {
#pragma unroll
                for (int k = 0; k < GK; ++k) cs += As[k * GT + ((((tid >> 2) ^ ((k & 1) << 2)) << 2) + (tid & 3))];
            }
        } else if (TA) {
#pragma unroll
            for (int s4 = 0; s4 < GK / 4; ++s4) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = As[(4 * s4 + q) * GT + colA[i]];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[(4 * s4 + q) * GT + colB[j]];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (do_sum && tid < GT) {
#pragma unroll
                for (int k = 0; k < GK; ++k) cs += As[k * GT + ((((tid >> 2) ^ ((k & 1) << 2)) << 2) + (tid & 3))];
            }
        } else {
            f32x4 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const f32x4 *>(As + colA[i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[(4 * q + r) * GT + colB[j]];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][r], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // one accessor for both accumulator layouts: value of row `row` (0..63 of this wave's block, as enumerated below) x column
    // f32 form: tile (i, j) register r  -> row 16 i + 4 q + r,                       column 16 j + c16
    // X3 form : tile (i, j) register e  -> row 32 i + (e & 3) + 8 (e >> 2) + 4 h,    column 32 j + r32
    constexpr int NI = X3 ? 2 : 4, NR = X3 ? 16 : 4;
    auto row_of = [&](int i, int e) { return X3 ? 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h : 16 * i + 4 * q + e; };
    auto col_of = [&](int j) { return X3 ? 32 * j + r32 : 16 * j + c16; };
    auto val_of = [&](int i, int j, int e) -> float { if constexpr (X3) return acx[i][j][e]; else return acc[i][j][e]; };
    if (!TA && mode != 2 && ((reinterpret_cast<uintptr_t>(C) | (mask ? reinterpret_cast<uintptr_t>(mask) : 0)) & 15) == 0 && ldc % 4 == 0 &&
        (mask == nullptr || ldm % 4 == 0)) {
        // Data-gradient epilogue through LDS: the accumulator layout gives 64-byte row segments per store instruction (64 mask loads
        // + 64 stores per lane); staged, every thread moves whole float4s of 512-byte rows (8 stores per lane per half).
        // Everything the stores need from memory -- the mask (as sign bits or as the fp32 activation) and, for mode 1, the old value -- is
        // fetched by ONE batch of loads per half through clamped addresses (no per-lane branches), ahead of the staging barrier; the
        // stores then follow each other with no wait between them.  (Round 3, from the cycle stamps of tools/stamp_gemm.py: with a load
        // inside each store's guard the compiler put `s_waitcnt vmcnt(0)` -- which on gfx9 also waits for every earlier STORE -- in front of
        // all 16 stores of a lane: a serial chain of 16 memory round trips, 60 % of this kernel's wave time.)
        constexpr int SLD = GT + 4;
        float *stage = &smem[0][0][0];                      // 64 x 132 floats = 33 KB of the 48 KB ring, free after the K loop
        const bool by_bits = mask != nullptr && bits != nullptr, by_mask = mask != nullptr && bits == nullptr, whole4 = (N & 3) == 0;
        const int erow = tid >> 5, en = n0 + 4 * (tid & 31);
        unsigned nibs[2] = {0u, 0u};                        // this lane's 16 mask nibbles (row erow + 8 e, columns en .. en+3), 4 bits each
        if (by_bits) {
#pragma unroll
            for (int e = 0; e < 16; ++e) nibs[e >> 3] |= (((unsigned)pre_raw[e] >> (en & 4)) & 15u) << (4 * (e & 7));
        }
        lds_barrier();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {             // the rarer fp32-mask / accumulate forms fetch in two batches of four per half
                f32x4 mk[4], od[4];
                if (whole4 && by_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long m = m0 + 64 * half + erow + 8 * (4 * grp + e);
                        mk[e] = *reinterpret_cast<const f32x4 *>(mask + ((m < M && en < N) ? m * ldm + en : 0));
                    }
                }
                if (whole4 && mode == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const long m = m0 + 64 * half + erow + 8 * (4 * grp + e);
                        od[e] = *reinterpret_cast<const f32x4 *>(C + ((m < M && en < N) ? m * ldc + en : 0));
                    }
                }
                if (grp == 0) {
                    if (wm == half) {
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
#pragma unroll
                                for (int r = 0; r < NR; ++r) stage[row_of(i, r) * SLD + 64 * wn + col_of(j)] = val_of(i, j, r);
                    }
                    lds_barrier();
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = erow + 8 * (4 * grp + e);
                    const long m = m0 + 64 * half + row;
                    const unsigned nb = nibs[half] >> (4 * (4 * grp + e));
                    f32x4 v = *reinterpret_cast<const f32x4 *>(stage + row * SLD + 4 * (tid & 31));
                    if (whole4) {
                        if (by_bits) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] *= ((nb >> k) & 1u) ? 1.0f : slope;
                        } else if (by_mask) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] *= (mk[e][k] > 0.0f) ? 1.0f : slope;
                        }
                        if (mode == 1) { v[0] += od[e][0]; v[1] += od[e][1]; v[2] += od[e][2]; v[3] += od[e][3]; }
                        if (m < M && en < N) *reinterpret_cast<f32x4 *>(C + m * ldc + en) = v;
                    } else if (m < M && en < N) {            // ragged N (the encodings' gradients): element-wise, rare and small
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (en + k < N) {
                                float x = v[k];
                                if (by_bits) x *= ((nb >> k) & 1u) ? 1.0f : slope;
                                else if (by_mask) x *= (mask[m * ldm + en + k] > 0.0f) ? 1.0f : slope;
                                float *dst = C + m * ldc + en + k;
                                *dst = (mode == 1) ? *dst + x : x;
                            }
                        }
                    }
                }
            }
            lds_barrier();
        }
        return;
    }
    if (TA && mode == 2) {
        // Weight-gradient epilogue through LDS as well: one atomic instruction then covers 256 contiguous bytes of a dW row
        // instead of four 64-byte segments of four rows.
        constexpr int SLD = GT + 4;
        float *stage = &smem[0][0][0];
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r = 0; r < NR; ++r) stage[row_of(i, r) * SLD + 64 * wn + col_of(j)] = val_of(i, j, r);
            }
            __syncthreads();
#pragma unroll 4
            for (int e = 0; e < 32; ++e) {
                const int idx = tid + 256 * e, row = idx >> 7, col = idx & 127;
                const long m = m0 + 64 * half + row;
                const int n = n0 + col;
                if (m < M && n < N && ldc > 0) atomicAdd(C + m * ldc + n, stage[row * SLD + col]);
            }
            __syncthreads();
        }
        if (do_sum && tid < GT && m0 + tid < M) atomicAdd(rowsum + m0 + tid, cs);
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const long m = m0 + 64 * wm + row_of(i, r);
                const int n = n0 + 64 * wn + col_of(j);
                if (m < M && n < N) {
                    float v = val_of(i, j, r);
                    if (mask != nullptr) v *= (mask[m * ldm + n] > 0.0f) ? 1.0f : slope;
                    float *dst = C + m * ldc + n;
                    if (mode == 0) *dst = v;
                    else if (mode == 1) *dst += v;
                    else atomicAdd(dst, v);
                }
            }
    if (do_sum && tid < GT && m0 + tid < M) atomicAdd(rowsum + m0 + tid, cs);
}

// ------------------------------------------------------------------------------------------------
// The weight-gradient GEMM on the bf16 pipe, operands split ONCE per workgroup (round 3).  gemm_dma_kernel<true, true> splits every fp32
// operand value on its way from LDS into the MFMA -- in each of the two waves that use it, behind eight scalar LDS reads per fragment --
// and its K-step body (286 instructions per wave for 12 MFMAs) is what bounds it: s_memtime stamps (tools/stamp_gemm.py) put 2,250 of a
// K-step's 2,530 cycles into the body, 120 into the wait for the LDS-DMA and 155 into the barrier.  Here a K-step is two phases:
//   split   thread (operand = tid >> 7, column = tid & 127) reads its column's 16 samples of the fp32 stage (conflict-free: a wave reads 64
//           consecutive columns of one sample row), splits them (48 VALU instructions) and writes the column's hi and lo bf16 fragments
//           [column][16 k] with four ds_write_b128; the bias-gradient column sum rides on the same registers;
//   MFMA    a wave fetches its 2 + 2 (hi, lo) fragment pairs with eight ds_read_b128 (lane (r32, h): 16 bytes at column * 32 + 16 h) and
//           issues the 12 MFMAs.
// Same tiles, LDS-DMA ring (depth 4), XCD-aware 1-D grid, sign-bit emission, staged atomic epilogue and arguments as the kernel it replaces
// for X3; 64 KB ring + 16 KB fragments = 80 KB of LDS: two workgroups per CU.
constexpr int TN_DST = 4, TN_THREADS = 512;
constexpr int TN_RING_FLOATS = TN_DST * 2 * DTILE;
constexpr int TN_FRAG_DWORDS = 2 * 2 * GT * 8;             // [operand][hi | lo][column][8 dwords = 16 bf16]
constexpr int TN_LDS_BYTES = (TN_RING_FLOATS + TN_FRAG_DWORDS) * 4;
static_assert(TN_LDS_BYTES == 81920, "two workgroups per CU: 2 x 80 KB = the 160 KB of a gfx950 CU");

// One 128 x 128 tile of C += A^T B over the samples [k_lo, k_hi): the body shared by the per-layer kernel (one tile-slab per workgroup)
// and the job-table kernel (a workgroup walks many).  Ends on an LDS barrier: the ring and the fragment buffer are free on return.
// position (dwords) of column `col`'s 4 dwords (8 bf16: samples 8 h .. 8 h + 7 of a K-step) inside one [hi] or [lo] fragment block of GT
// columns: [32-column group][h][column of the group][4 dwords] -- the MFMA phase's lane (r32, h) and the split phase's thread (column, h) both
// touch 16 bytes at 16 x lane: conflict-free ds_read_b128 / ds_write_b128.  (Round 4, first form: [column][h][4 dwords] = a 32-byte lane
// stride, two lanes per bank group -- SQ_LDS_BANK_CONFLICT 1.5-1.7 x SQ_ACTIVE_INST_LDS in both weight-gradient kernels.)
__device__ __forceinline__ constexpr int frag_pos(int col, int h) { return ((col >> 5) * 64 + h * 32 + (col & 31)) * 4; }

__device__ __forceinline__ void tn_tile(float *tn_lds, int M, int N, const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                        float *__restrict__ C, long ldc, long k_lo, long k_hi, float *__restrict__ rowsum, int bx, int by,
                                        const float *__restrict__ zero, unsigned char *__restrict__ bits)
{
    float *ring = tn_lds;                                                    // [TN_DST][2][DTILE]
    uint32_t *frag = reinterpret_cast<uint32_t *>(tn_lds + TN_RING_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, h = lane >> 5, r32 = lane & 31, c32 = lane & 31;
    const long m0 = (long)by * GT;
    const int n0 = bx * GT;
    const int T = (int)((k_hi - k_lo + GK - 1) / GK);
    const bool do_sum = rowsum != nullptr && bx == 0;

    // this wave's two DMA instructions per K-step: waves 0..3 move the dY tile, waves 4..7 the X tile; K-major [16 k][128 cols], 32 chunks
    // per row, two rows per instruction, XOR-swizzled by 16 floats on odd rows (kept from gemm_dma_kernel: the sign-bit pass relies on it)
    const float *src[2]; long step[2]; int krow[2]; bool colok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rp = (wave & 3) * 2 + u;
        const float *base = (wave < 4) ? A : B;
        const long ld = (wave < 4) ? lda : ldb;
        const long c0 = (wave < 4) ? m0 : n0;
        const int dim = (wave < 4) ? M : N;
        const int row = 2 * rp + h;
        const int j = c32 ^ ((row & 1) << 2);
        src[u] = base + (k_lo + row) * ld + c0 + 4 * j;
        step[u] = GK * ld;
        krow[u] = row;
        colok[u] = c0 + 4 * j < dim;
    }
    const int dst_off = ((wave < 4) ? 0 : DTILE) + (wave & 3) * 2 * 256;
    auto issue = [&](int t) {
        float *dstb = ring + (t % TN_DST) * 2 * DTILE + dst_off;
        const long k0 = k_lo + (long)t * GK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float *g = (colok[u] && k0 + krow[u] < k_hi) ? src[u] + (long)t * step[u] : zero;
#if defined(SAHS_DIAG) && defined(SAHS_TN_NODMA)      // timing-only (results wrong by construction): the K loop without its operand fetch
            asm volatile("" :: "v"(g), "v"(dstb));
#else
            __builtin_amdgcn_global_load_lds((gbl_void_t)g, (lds_void_t)(dstb + u * 256), 16, 0, 0);
#endif
        }
    };
    // split phase: this thread's column of its operand tile, at its two swizzle positions (even / odd sample rows)
    const int sop = tid >> 8, skh = (tid >> 7) & 1, scol = tid & 127;        // samples 8 skh .. 8 skh + 7 of the column
    const int spos0 = scol, spos1 = (((scol >> 2) ^ 4) << 2) + (scol & 3);
    uint32_t *fdst = frag + (sop * 2) * GT * 8 + frag_pos(scol, skh);        // hi block of this column; the lo block is GT * 8 dwords further
    // MFMA phase: fragment rows of this wave's 2 + 2 column blocks
    const uint32_t *fa[2], *fb;                                              // wave (wm, wn): rows 64 wm + 32 i, columns 32 wn
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = frag + (0 * 2) * GT * 8 + frag_pos(64 * wm + 32 * i + r32, h);
    fb = frag + (1 * 2) * GT * 8 + frag_pos(32 * wn + r32, h);
    f32x16_t acx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acx[i][e] = 0.0f;
    float cs = 0.0f;

    // vmcnt accounting as in gemm_dma_kernel: the wave's only loads are its 2 LDS-DMA instructions per issue(), completing in issue order;
    // anything older still in flight (the sign-bit stores, a previous tile's atomics) can only make a counted wait stricter
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    for (int t = 0; t < T; ++t) {
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // (every wave has finished the MFMA phase of step t-1: the fragment buffer and ring stage (t-1) % 4 are free)
        if (t + TN_DST - 1 < T) issue(t + TN_DST - 1);
        const float *stage = ring + (t % TN_DST) * 2 * DTILE;
        {
            const float *col = stage + sop * DTILE + 8 * skh * GT;
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = col[k * GT + ((k & 1) ? spos1 : spos0)];
            u32x4_t hi, lo;
            split8(x, hi, lo);
            *reinterpret_cast<u32x4_t *>(fdst) = hi;
            *reinterpret_cast<u32x4_t *>(fdst + GT * 8) = lo;
            if (do_sum && sop == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) cs += x[k];
            }
        }
        if (bits != nullptr && by == 0 && tid < 256) {      // thread (kk, g): the 8 columns 8g .. 8g+7 of sample row kk of the X tile -> one byte
            const int kk = tid >> 4, g = tid & 15;
            const long k = k_lo + (long)t * GK + kk;
            if (k < k_hi && n0 + 8 * g < N) {
                const float *px = stage + DTILE + kk * GT + 4 * ((2 * g) ^ ((kk & 1) << 2));     // chunks 2g, 2g+1 stay adjacent under the swizzle
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(px), v1 = *reinterpret_cast<const f32x4 *>(px + 4);
                unsigned b8 = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) b8 |= (v0[i] > 0.0f ? 1u : 0u) << i | (v1[i] > 0.0f ? 1u : 0u) << (4 + i);
                bits[k * (N >> 3) + (n0 >> 3) + g] = (unsigned char)b8;
            }
        }
        lds_barrier();
        u32x4_t ah[2], al[2], bh, bl;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const u32x4_t *>(fa[i]);
            al[i] = *reinterpret_cast<const u32x4_t *>(fa[i] + GT * 8);
        }
        bh = *reinterpret_cast<const u32x4_t *>(fb);
        bl = *reinterpret_cast<const u32x4_t *>(fb + GT * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) acx[i] = mfma3(ah[i], al[i], bh, bl, acx[i]);
    }
    // epilogue through LDS: one atomic instruction covers 256 contiguous bytes of a dW row.  Accumulator tile i register e is
    // row 32 i + (e & 3) + 8 (e >> 2) + 4 h, column r32 of this wave's 64 x 32 block
    constexpr int SLD = GT + 4;
    float *out = ring;                                   // 64 x 132 floats = 33 KB of the ring, free after the K loop
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) out[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * SLD + 32 * wn + r32] = acx[i][e];
        }
        lds_barrier();
#pragma unroll 4
        for (int e = 0; e < 16; ++e) {
            const int idx = tid + TN_THREADS * e, row = idx >> 7, col = idx & 127;
            const long m = m0 + 64 * half + row;
            const int n = n0 + col;
            if (m < M && n < N && ldc > 0) atomicAdd(C + m * ldc + n, out[row * SLD + col]);
        }
        lds_barrier();
    }
    if (do_sum && sop == 0 && m0 + scol < M) atomicAdd(rowsum + m0 + scol, cs);      // (two partial sums per column: samples 0..7 and 8..15 of every step)
}

__global__ void __launch_bounds__(TN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) gemm_tn_split_kernel(int M, int N, int K, const float *__restrict__ A, long lda,
                                                       const float *__restrict__ B, long ldb, float *__restrict__ C, long ldc, int kslab,
                                                       float *__restrict__ rowsum, int nx, const float *__restrict__ zero,
                                                       unsigned char *__restrict__ bits)
{
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    // 1-D grid: the nx x ny output tiles of ONE sample slab read the same dY and X rows -- they run on the same XCD (workgroup ids congruent
    // mod 8, consecutive slots), so that XCD's L2 fetches the slab from HBM once instead of once per tile
    const int ny = (M + GT - 1) / GT, tiles = nx * ny;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, tile = slot % tiles;
    const long bz = (long)(slot / tiles) * 8 + xcd;
    if (bz * kslab >= K) return;
    const long k_lo = bz * kslab;
    const long k_hi = (k_lo + kslab < K) ? k_lo + kslab : K;
    tn_tile(tn_lds, M, N, A, lda, B, ldb, C, ldc, k_lo, k_hi, rowsum, tile % nx, tile / nx, zero, bits);
}

// ------------------------------------------------------------------------------------------------
// ALL weight-gradient GEMMs of a walk in ONE launch (round 4).  Per launch the kernel above pays ring fill, a K loop with one workgroup
// per CU, 64 KB of atomics per workgroup, ramp and tail -- 35-40 us of a 45-200 us launch, 82 times per training step (DESIGN.md section
// 7).  Here persistent workgroups (two per CU) walk a job table: job = one layer's dW (+ db) = dY^T X with dY, X dense [P x width]
// planes; item = (sample range r, job, 128 x 128 tile), ranges of `range` samples (the same for every job: items cost the same), item
// index = r * tiles_total + tile, so that the tiles of one job and range -- which stream the same dY and X rows -- sit on neighbouring
// workgroups of ONE XCD (virtual id below) at the same time and share that L2's fetches.  A range is ~10 k samples instead of the 512 of
// a slab: 20 x fewer atomic epilogues.  The table travels as kernel arguments (no upload, nothing allocated).
struct TnJob { const float *A; const float *B; float *C; float *rowsum; long lda, ldb, ldc; int M, N; };
constexpr int MAX_TN_JOBS = 40;
struct TnBatch { TnJob j[MAX_TN_JOBS]; };
static_assert(sizeof(TnBatch) <= 3072, "kernel-argument budget");

__global__ void __launch_bounds__(TN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) gemm_tn_jobs_kernel(TnBatch jobs, int njobs, int tiles_total, long P, long range,
                                                                                                             const float *__restrict__ zero)
{
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    const int G = gridDim.x;                                                  // a multiple of 8
    const int vid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);          // consecutive virtual ids = consecutive slots of one XCD
    const long nrange = (P + range - 1) / range;
    const long items = nrange * tiles_total;
    for (long it = vid; it < items; it += G) {
        const long r = it / tiles_total;
        int tg = (int)(it - r * tiles_total), j = 0;
        for (; j < njobs - 1; ++j) {
            const int tj = ((jobs.j[j].N + GT - 1) / GT) * ((jobs.j[j].M + GT - 1) / GT);
            if (tg < tj) break;
            tg -= tj;
        }
        const TnJob &J = jobs.j[j];
        const int nx = (J.N + GT - 1) / GT;
        const long k_lo = r * range, k_hi = (k_lo + range < P) ? k_lo + range : P;
        tn_tile(tn_lds, J.M, J.N, J.A, J.lda, J.B, J.ldb, J.C, J.ldc, k_lo, k_hi, J.rowsum, tg % nx, tg / nx, zero, nullptr);
    }
}


// ------------------------------------------------------------------------------------------------
// The wide layers' weight gradients as 256 x 256 BLOCKS (round 4).  With 128 x 128 tiles a 256 x 256 layer is four workgroups that each
// stream 128 columns of dY and 128 of X: every operand byte is fetched twice unless the four happen to run in step on one XCD (measured on
// the training step: 21 GB fetched per step for 15.9 GB of operands, at 5 TB/s -- the launch is HBM-bound, so the re-reads are its time).
// Here ONE workgroup owns the whole 256 x 256 block of a job over a sample range: per 16-sample K-step it brings in 16 x 256 of dY and
// 16 x 256 of X once (32 KB by LDS-DMA, ring of four), splits both into bf16 hi / lo fragments once, and its eight waves (4 along M x 2
// along N, a 64 x 128 sub-block = eight 32 x 32 accumulators each) issue 24 MFMAs per wave and K-step.  160 KB of LDS: one workgroup per
// CU, two waves per SIMD.  The accumulators go out as atomics straight from the registers: one register of a 32 x 32 accumulator is two
// 128-byte row segments, which the memory side takes at full rate (MI355X_MICROARCH.md, global float atomics) -- no staging pass.
// Jobs: M, N <= 256 (columns past M / N come from the zero page); the narrower jobs stay with the 128 x 128 kernel above.
constexpr int TW_DST = 3, TW_UNITS = 4;                                       // ring depth; 128-column operand blocks per K-step (2 of dY + 2 of X)
constexpr int TW_STAGE_FLOATS = TW_UNITS * DTILE;                              // 8192 floats = 32 KB
constexpr int TW_RING_FLOATS = TW_DST * TW_STAGE_FLOATS;
constexpr int TW_FRAG_DWORDS = TW_UNITS * 2 * GT * 8;                          // [block][hi | lo][frag_pos(column, h)]: 32 KB, two of them
constexpr int TW_LDS_BYTES = (TW_RING_FLOATS + 2 * TW_FRAG_DWORDS) * 4;
static_assert(TW_LDS_BYTES == 163840, "one workgroup per CU: all 160 KB of it");

// Software-pipelined: while the MFMAs of K-step t run from fragment buffer t & 1, the same waves split K-step t + 1 from the ring into the
// other fragment buffer (VALU and LDS work beside the matrix pipe instead of in turns with it: in turns the kernel was instruction-bound at
// 2 us per K-step, 3.7 TB/s), ONE barrier per K-step, two K-steps of LDS-DMA in flight behind the one being split.
__device__ __forceinline__ void tn_block256(float *tn_lds, int M, int N, const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                            float *__restrict__ C, long ldc, long k_lo, long k_hi, float *__restrict__ rowsum,
                                            const float *__restrict__ zero)
{
    float *ring = tn_lds;                                                      // [TW_DST][block 0..3][16 k][128 cols]; blocks 0, 1 = dY, 2, 3 = X
    uint32_t *frag = reinterpret_cast<uint32_t *>(tn_lds + TW_RING_FLOATS);   // [2][TW_FRAG_DWORDS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5, r32 = lane & 31, c32 = lane & 31;
    const int T = (int)((k_hi - k_lo + GK - 1) / GK);
    // DMA: 32 one-KB units per K-step (block b, row pair rp); wave w moves units 4 w .. 4 w + 3 = row pairs 4 (w & 1) .. of block w >> 1.
    // K-major [16 k][128 cols], XOR-swizzled by 16 floats on odd rows as in tn_tile.
    const float *src[4]; long step[4]; int krow[4]; bool colok[4];
    const int blk = wave >> 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rp = (wave & 1) * 4 + u;
        const float *base = (blk < 2) ? A : B;
        const long ld = (blk < 2) ? lda : ldb;
        const int c0 = (blk & 1) * GT;
        const int dim = (blk < 2) ? M : N;
        const int row = 2 * rp + h;
        const int j = c32 ^ ((row & 1) << 2);
        src[u] = base + (k_lo + row) * ld + c0 + 4 * j;
        step[u] = GK * ld;
        krow[u] = row;
        colok[u] = c0 + 4 * j < dim;
    }
    const int dst_off = blk * DTILE + (wave & 1) * 4 * 256;
    auto issue = [&](int t) {
        float *dstb = ring + (t % TW_DST) * TW_STAGE_FLOATS + dst_off;
        const long k0 = k_lo + (long)t * GK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *g = (colok[u] && k0 + krow[u] < k_hi) ? src[u] + (long)t * step[u] : zero;
#if defined(SAHS_DIAG) && defined(SAHS_TN_NODMA)      // timing-only (results wrong by construction): the K loop without its operand fetch
            asm volatile("" :: "v"(g), "v"(dstb));
#else
            __builtin_amdgcn_global_load_lds((gbl_void_t)g, (lds_void_t)(dstb + u * 256), 16, 0, 0);
#endif
        }
    };
    // split, two rounds per K-step: round r, thread -> block 2 r + (tid >> 8), samples 8 skh .. 8 skh + 7 of column scol (round 0: the dY blocks)
    const int sb = tid >> 8, skh = (tid >> 7) & 1, scol = tid & 127;
    const int spos0 = scol, spos1 = (((scol >> 2) ^ 4) << 2) + (scol & 3);
    float cs = 0.0f;
    auto split_round = [&](int t, int r) {
        const float *col = ring + (t % TW_DST) * TW_STAGE_FLOATS + (2 * r + sb) * DTILE + 8 * skh * GT;
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = col[k * GT + ((k & 1) ? spos1 : spos0)];
        u32x4_t hi, lo;
        split8(x, hi, lo);
        uint32_t *fdst = frag + (t & 1) * TW_FRAG_DWORDS + ((2 * r + sb) * 2) * GT * 8 + frag_pos(scol, skh);
        *reinterpret_cast<u32x4_t *>(fdst) = hi;
        *reinterpret_cast<u32x4_t *>(fdst + GT * 8) = lo;
        if (r == 0 && rowsum != nullptr) {
#pragma unroll
            for (int k = 0; k < 8; ++k) cs += x[k];
        }
    };
    // MFMA: wave (wm, wn) owns rows 64 wm .. + 63 (dY block wm >> 1), columns 128 wn .. + 127 (X block wn)
    int fa[2], fb[4];      // dword offsets inside a fragment buffer
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = ((wm >> 1) * 2) * GT * 8 + frag_pos(64 * (wm & 1) + 32 * i + r32, h);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = ((2 + wn) * 2) * GT * 8 + frag_pos(32 * j + r32, h);
    f32x16_t acx[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acx[i][j][e] = 0.0f;

    // vmcnt accounting as in tn_tile: this wave's loads are its 4 LDS-DMA instructions per issue(), in issue order; anything older still in
    // flight (a previous item's atomics) can only make a counted wait stricter
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    if (T > 0) {      // K-step 0 has landed when at most the 8 DMA instructions of steps 1, 2 are outstanding
        if (T > 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (T > 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        split_round(0, 0);
        split_round(0, 1);
    }
    for (int t = 0; t < T; ++t) {
        // K-step t + 1 has landed (t + 2 may be in flight); every wave has finished the MFMAs of step t - 1 and the split of step t
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 3 < T) issue(t + 3);                         // into stage t % 3, which the split of step t (last iteration) was the last to read
        const uint32_t *fr = frag + (t & 1) * TW_FRAG_DWORDS;
        u32x4_t ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const u32x4_t *>(fr + fa[i]);
            al[i] = *reinterpret_cast<const u32x4_t *>(fr + fa[i] + GT * 8);
        }
        // every fragment of the K-step is requested before the first MFMA (the compiler otherwise reads a column block's pair right in front of
        // its six MFMAs and waits: four exposed LDS round trips per K-step with two waves per SIMD to cover them)
        u32x4_t bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const u32x4_t *>(fr + fb[j]);
            bl[j] = *reinterpret_cast<const u32x4_t *>(fr + fb[j] + GT * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acx[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acx[i][j]);
            if (t + 1 < T && (j & 1) == 0) split_round(t + 1, j >> 1);      // beside the MFMAs: the next K-step's fragments
        }
    }
    // accumulator (i, j) register e: row 64 wm + 32 i + (e & 3) + 8 (e >> 2) + 4 h, column 128 wn + 32 j + r32
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 128 * wn + 32 * j + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) atomicAdd(C + (long)m * ldc + n, acx[i][j][e]);
            }
        }
    if (rowsum != nullptr && sb * GT + scol < M) atomicAdd(rowsum + sb * GT + scol, cs);      // (two partial sums per column: samples 0..7 and 8..15 of every step)
    __syncthreads();      // the ring and the fragment buffers are free (and this item's DMA has long drained) before the next item issues into them
}

__global__ void __launch_bounds__(TN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_tn_jobs256_kernel(TnBatch jobs, int njobs, long P, long range,
                                                                                                               const float *__restrict__ zero)
{
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    const long nrange = (P + range - 1) / range;
    const long items = nrange * njobs;                                        // item = (range r, job): the jobs of one range side by side
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const long r = it / njobs;
        const TnJob &J = jobs.j[(int)(it - r * njobs)];
        const long k_lo = r * range, k_hi = (k_lo + range < P) ? k_lo + range : P;
        tn_block256(tn_lds, J.M, J.N, J.A, J.lda, J.B, J.ldb, J.C, J.ldc, k_lo, k_hi, J.rowsum, zero);
    }
}

// ------------------------------------------------------------------------------------------------
// The same two job-table launches in exact fp32 products (ops.backward_gemm_precision("fp32"): the reference's arithmetic), on
// v_mfma_f32_16x16x4_f32.  Same operand tiles ([16 k][128 cols] K-major, XOR-swizzled on odd rows) brought in by the same LDS-DMA ring; no
// split phase: the MFMA lanes read their operands straight from the ring (lane (q, c16): sample 4 s4 + q, column 16 i + c16 -- ds_read_b32,
// conflict-free under the swizzle), four samples per MFMA.  Bound: the f32 matrix pipe -- a 256 x 256 block is 128 MFMAs per wave and K-step
// (4096 cycles) against 32 KB of operands -- not HBM.  Ring of four K-steps.
constexpr int TF_DST = 4;
constexpr int TF_LDS_BYTES = TF_DST * 2 * DTILE * 4;                           // 64 KB: two workgroups per CU
constexpr int TFW_DST = 5;                                                     // the wide kernel's ring: all 160 KB of the CU, four K-steps in flight
constexpr int TFW_LDS_BYTES = TFW_DST * TW_STAGE_FLOATS * 4;                   // 160 KB: one workgroup per CU
__device__ __forceinline__ int swz_col(int col, int odd) { return (((col >> 2) ^ (odd << 2)) << 2) + (col & 3); }

// One (up to) 128 x 128 tile of C += A^T B over the samples [k_lo, k_hi).  The tile's valid part is TR x TC 16 x 16 accumulators (a head's
// dW is 1 x 8, a 64-wide layer against PE(x) 4 x 4, a trunk layer against PE(w) 8 x 2): the eight waves are laid over it as RW x CW with
// CW = 4, 2, 1 for TC > 4, > 2, <= 2, a wave owning up to 4 x 2 accumulators -- only the valid ones are multiplied (wave-uniform bounds),
// and the work stays spread evenly over the four SIMDs (wave & 3).
__device__ __forceinline__ void tn_tile_f32(float *ring, int M, int N, const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                            float *__restrict__ C, long ldc, long k_lo, long k_hi, float *__restrict__ rowsum, int bx, int by,
                                            const float *__restrict__ zero)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31, q = lane >> 4, c16 = lane & 15;
    const long m0 = (long)by * GT;
    const int n0 = bx * GT;
    const int T = (int)((k_hi - k_lo + GK - 1) / GK);
    const bool do_sum = rowsum != nullptr && bx == 0;
    const int mt = (int)(M - m0 < GT ? M - m0 : GT), nt = N - n0 < GT ? N - n0 : GT;
    const int TR = (mt + 15) >> 4, TC = (nt + 15) >> 4;
    const int CW = TC > 4 ? 4 : (TC > 2 ? 2 : 1), RW = 8 / CW, RPW = (TR + RW - 1) / RW;      // RPW <= 4
    const int wm = wave / CW, wn = wave % CW;
    const int rt0 = wm * RPW, ct0 = 2 * wn;
    const int ni = __builtin_amdgcn_readfirstlane(TR - rt0 < 0 ? 0 : (TR - rt0 < RPW ? TR - rt0 : RPW));
    const int nj = __builtin_amdgcn_readfirstlane(TC - ct0 < 0 ? 0 : (TC - ct0 < 2 ? TC - ct0 : 2));
    // DMA as in tn_tile: waves 0..3 move the dY tile, waves 4..7 the X tile, two instructions per wave and K-step
    // (K-steps are issued in order: the lane's source pointers simply advance; columns past the operand's width read the zero page and stay
    // there; rows past k_hi exist in the last K-step of a ragged range only -- a uniform test)
    const float *cur[2]; long step[2]; int krow[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rp = (wave & 3) * 2 + u;
        const float *base = (wave < 4) ? A : B;
        const long ld = (wave < 4) ? lda : ldb;
        const long c0 = (wave < 4) ? m0 : n0;
        const int dim = (wave < 4) ? M : N;
        const int row = 2 * rp + h;
        const int j = c32 ^ ((row & 1) << 2);
        const bool colok = c0 + 4 * j < dim;
        cur[u] = colok ? base + (k_lo + row) * ld + c0 + 4 * j : zero;
        step[u] = colok ? GK * ld : 0;
        krow[u] = row;
    }
    const int dst_off = ((wave < 4) ? 0 : DTILE) + (wave & 3) * 2 * 256;
    auto issue = [&](int t) {
        float *dstb = ring + (t % TF_DST) * 2 * DTILE + dst_off;
        const long k0 = k_lo + (long)t * GK;
        const bool tail = k0 + GK > k_hi;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float *g = (tail && k0 + krow[u] >= k_hi) ? zero : cur[u];
#if defined(SAHS_DIAG) && defined(SAHS_TNF_NODMA)      // timing-only (results wrong by construction): the K loop without its operand fetch
            asm volatile("" :: "v"(g), "v"(dstb));
#else
            __builtin_amdgcn_global_load_lds((gbl_void_t)g, (lds_void_t)(dstb + u * 256), 16, 0, 0);
#endif
            cur[u] += step[u];
        }
    };
    int colA[4], colB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) colA[i] = swz_col((16 * (rt0 + i) + c16) & (GT - 1), q & 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) colB[j] = DTILE + swz_col((16 * (ct0 + j) + c16) & (GT - 1), q & 1);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float cs = 0.0f;
    // vmcnt accounting as in tn_tile: the wave's only loads are its 2 LDS-DMA instructions per issue(), completing in issue order
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    // The K loop, once per shape of the wave block (4, 2 or 1 row tiles by 2 column tiles; NI = 0: any other -- ragged widths -- with a uniform
    // guard per accumulator): the dispatch is outside the loop, so the loop body is MFMAs and operand reads only (with the guards inside, the
    // compiler put a vcc branch between any two of them)
    auto kloop = [&]<int NI, int NJ>() {
        for (int t = 0; t < T; ++t) {
            if (t + 2 < T) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const float *stage = ring + (t % TF_DST) * 2 * DTILE;
            // K-step t + 3 goes into the stage every wave finished reading before the barrier above -- requested behind the first quarter of
            // this step's MFMAs, not in front of them: its address arithmetic then runs beside the matrix pipe instead of holding it up
            const bool work = NI > 0 || (ni > 0 && nj > 0);
            if (!work && t + TF_DST - 1 < T) issue(t + TF_DST - 1);
            if (work) {
#pragma unroll
                for (int s4 = 0; s4 < GK / 4; ++s4) {
                    if (s4 == 1 && t + TF_DST - 1 < T) issue(t + TF_DST - 1);
                    const float *row = stage + (4 * s4 + q) * GT;
                    if constexpr (NI > 0) {
                        float a[NI], b[NJ];
#pragma unroll
                        for (int i = 0; i < NI; ++i) a[i] = row[colA[i]];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) b[j] = row[colB[j]];
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                if (i < ni && j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[colA[i]], row[colB[j]], acc[i][j], 0, 0, 0);
                    }
                }
            }
            if (do_sum && tid < GT) {      // (measured: the launch is as long without these sums, 0.650 vs 0.652 ms)
#pragma unroll
                for (int k = 0; k < GK; ++k) cs += stage[k * GT + swz_col(tid, k & 1)];
            }
        }
    };
    if (nj == 2 && ni == 4) kloop.template operator()<4, 2>();
    else if (nj == 2 && ni == 2) kloop.template operator()<2, 2>();
    else if (nj == 2 && ni == 1) kloop.template operator()<1, 2>();
    else kloop.template operator()<0, 0>();
    // accumulator (i, j) register r: row 16 (rt0 + i) + 4 q + r, column 16 (ct0 + j) + c16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (i < ni && j < nj) {
                const int n = n0 + 16 * (ct0 + j) + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long m = m0 + 16 * (rt0 + i) + 4 * q + r;
                    if (m < M && n < N && ldc > 0) atomicAdd(C + m * ldc + n, acc[i][j][r]);
                }
            }
        }
    if (do_sum && tid < GT && m0 + tid < M) atomicAdd(rowsum + m0 + tid, cs);
    __syncthreads();      // the ring is free (this item's DMA has drained) before the next item issues into it
}

// The item table of an fp32 job launch.  Unit = one output tile (narrow kernel) or one 256 x 256 job (wide kernel); a unit's samples are
// cut into n ranges in proportion to what a K-step of it costs (a 16 x 128 head tile is 1/8 of the MFMAs of a 128 x 128 tile, a 128 x 256
// layer half of a 256 x 256 one), so that items cost the same and every workgroup gets the same number of them: with equal ranges the
// launch lasts as long as its dearest unit (measured: wide kernel 2.83 ms per 262,144 samples for 2.24 ms of matrix work in its longest item).
struct TnUnit { unsigned short start, n; unsigned char job, bx, by, pad; };
constexpr int MAX_TN_UNITS = 64;
struct TnPlan { TnUnit u[MAX_TN_UNITS]; int nunits, items; };
static_assert(sizeof(TnBatch) + sizeof(TnPlan) <= 3800, "kernel-argument budget");

__global__ void __launch_bounds__(TN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) gemm_tn_jobs_f32_kernel(TnBatch jobs, TnPlan plan, long P, const float *__restrict__ zero)
{
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    const int G = gridDim.x;
    for (int it = blockIdx.x; it < plan.items; it += G) {
        int u = 0;
        while (u + 1 < plan.nunits && (int)plan.u[u + 1].start <= it) ++u;
        const TnUnit U = plan.u[u];
        const long range = ((P + U.n - 1) / U.n + 15) / 16 * 16;
        const long k_lo = (long)(it - U.start) * range, k_hi = (k_lo + range < P) ? k_lo + range : P;
        if (k_lo >= P) continue;
        const TnJob &J = jobs.j[U.job];
        tn_tile_f32(tn_lds, J.M, J.N, J.A, J.lda, J.B, J.ldb, J.C, J.ldc, k_lo, k_hi, J.rowsum, U.bx, U.by, zero);
    }
}

// One whole 256 x 256 block: wave (wm, wn) of eight owns rows 64 wm .. (dY block wm >> 1), columns 128 wn .. (X block wn): 4 x 8 accumulators
__device__ __forceinline__ void tn_block256_f32(float *ring, int M, int N, const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                                float *__restrict__ C, long ldc, long k_lo, long k_hi, float *__restrict__ rowsum,
                                                const float *__restrict__ zero)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1, h = lane >> 5, c32 = lane & 31, q = lane >> 4, c16 = lane & 15;
    const int T = (int)((k_hi - k_lo + GK - 1) / GK);
    // DMA as in tn_block256: 32 one-KB units per K-step (block b, row pair rp); wave w moves row pairs 4 (w & 1) .. of block w >> 1
    // (pointers advance per K-step as in tn_tile_f32)
    const float *cur[4]; long step[4]; int krow[4];
    const int blk = wave >> 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rp = (wave & 1) * 4 + u;
        const float *base = (blk < 2) ? A : B;
        const long ld = (blk < 2) ? lda : ldb;
        const int c0 = (blk & 1) * GT;
        const int dim = (blk < 2) ? M : N;
        const int row = 2 * rp + h;
        const int j = c32 ^ ((row & 1) << 2);
        const bool colok = c0 + 4 * j < dim;
        cur[u] = colok ? base + (k_lo + row) * ld + c0 + 4 * j : zero;
        step[u] = colok ? GK * ld : 0;
        krow[u] = row;
    }
    const int dst_off = blk * DTILE + (wave & 1) * 4 * 256;
    auto issue = [&](int t) {
        float *dstb = ring + (t % TFW_DST) * TW_STAGE_FLOATS + dst_off;
        const long k0 = k_lo + (long)t * GK;
        const bool tail = k0 + GK > k_hi;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *g = (tail && k0 + krow[u] >= k_hi) ? zero : cur[u];
#if defined(SAHS_DIAG) && defined(SAHS_TNF_NODMA)      // timing-only (results wrong by construction): the K loop without its operand fetch
            asm volatile("" :: "v"(g), "v"(dstb));
#else
            __builtin_amdgcn_global_load_lds((gbl_void_t)g, (lds_void_t)(dstb + u * 256), 16, 0, 0);
#endif
            cur[u] += step[u];
        }
    };
    int colA[4], colB[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) colA[i] = (wm >> 1) * DTILE + swz_col(64 * (wm & 1) + 16 * i + c16, q & 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) colB[j] = (2 + wn) * DTILE + swz_col(16 * j + c16, q & 1);
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float cs = 0.0f;
    const int scol = (tid >> 7) * DTILE, sc = tid & 127;      // bias gradient: thread tid < 256 sums column tid of dY
    // vmcnt accounting: this wave's loads are its 4 LDS-DMA instructions per issue(), in issue order; anything older still in flight (a
    // previous item's atomics) can only make a counted wait stricter
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    if (T > 3) issue(3);
    for (int t = 0; t < T; ++t) {
        if (t + 3 < T) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else if (t + 2 < T) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const float *stage = ring + (t % TFW_DST) * TW_STAGE_FLOATS;
        const bool rows_valid = 64 * wm < M;      // (a 128 x 256 layer: waves 4..7 -- the second wave of every SIMD -- own rows of the zero page)
        if (!rows_valid && t + TFW_DST - 1 < T) issue(t + TFW_DST - 1);
        if (rows_valid) {
            // operands of sample group s4 + 1 are requested before the 32 MFMAs of group s4 (two register sets).  Measured MFMA busy 0.71 at
            // 2.35 GHz -- and the same launch time with the compiler's own read placement, with all four groups read up front and every
            // accumulator taking its four MFMAs in a row (the forward's pattern), with the DMA issue in front of the MFMAs, and without the
            // operand fetch (-7 %): LAB_NOTES R4.6
            float a[2][4], b[2][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[0][i] = stage[q * GT + colA[i]];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[0][j] = stage[q * GT + colB[j]];
#pragma unroll
            for (int s4 = 0; s4 < GK / 4; ++s4) {
                if (s4 + 1 < GK / 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[(s4 + 1) & 1][i] = stage[(4 * (s4 + 1) + q) * GT + colA[i]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[(s4 + 1) & 1][j] = stage[(4 * (s4 + 1) + q) * GT + colB[j]];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s4 == 1 && t + TFW_DST - 1 < T) issue(t + TFW_DST - 1);      // (behind the first quarter of the step's MFMAs: see tn_tile_f32)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#if defined(SAHS_DIAG) && defined(SAHS_TNF_NOMFMA)      // timing-only (results wrong by construction): the K loop without its matrix work
                        acc[i][j][0] += a[s4 & 1][i] * b[s4 & 1][j];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4 & 1][i], b[s4 & 1][j], acc[i][j], 0, 0, 0);
#endif
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (rowsum != nullptr && tid < 2 * GT) {
#pragma unroll
            for (int k = 0; k < GK; ++k) cs += stage[scol + k * GT + swz_col(sc, k & 1)];
        }
    }
    // accumulator (i, j) register r: row 64 wm + 16 i + 4 q + r, column 128 wn + 16 j + c16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = 128 * wn + 16 * j + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 64 * wm + 16 * i + 4 * q + r;
#if defined(SAHS_DIAG) && defined(SAHS_TNF_NOATOMIC)      // timing-only (results wrong by construction): the launch without its atomic epilogue
                if (64 * wm < M && m < M && n < N && acc[i][j][r] == 123.456f) C[(long)m * ldc + n] = 0.0f;
#else
                if (64 * wm < M && m < M && n < N) atomicAdd(C + (long)m * ldc + n, acc[i][j][r]);
#endif
            }
        }
    if (rowsum != nullptr && tid < 2 * GT && tid < M) atomicAdd(rowsum + tid, cs);
    __syncthreads();
}

__global__ void __launch_bounds__(TN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_tn_jobs256_f32_kernel(TnBatch jobs, TnPlan plan, long P, const float *__restrict__ zero)
{
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    for (int it = blockIdx.x; it < plan.items; it += gridDim.x) {
        int u = 0;
        while (u + 1 < plan.nunits && (int)plan.u[u + 1].start <= it) ++u;
        const TnUnit U = plan.u[u];
        const long range = ((P + U.n - 1) / U.n + 15) / 16 * 16;
        const long k_lo = (long)(it - U.start) * range, k_hi = (k_lo + range < P) ? k_lo + range : P;
        if (k_lo >= P) continue;
        const TnJob &J = jobs.j[U.job];
        tn_block256_f32(tn_lds, J.M, J.N, J.A, J.lda, J.B, J.ldb, J.C, J.ldc, k_lo, k_hi, J.rowsum, zero);
    }
}

// dst[m*ldd + n] (op)= src[m*lds + n] for n < N   (mode 0 copy, 1 add)
__global__ void copy2d_kernel(long M, int N, const float *__restrict__ src, long lds_, float *__restrict__ dst, long ldd, int mode)
{
    const long total = M * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long m = e / N; const int n = (int)(e % N);
        const float v = src[m * lds_ + n];
        if (mode) dst[m * ldd + n] += v; else dst[m * ldd + n] = v;
    }
}

// d_feat[m][:] += d_sigma[m] * w_alpha[:]   (rank-1: fc_alpha has one output row)
__global__ void rank1_add_kernel(long M, int N, const float *__restrict__ dcol, long ldc, const float *__restrict__ w, float *__restrict__ dst, long ldd)
{
    const long total = M * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long m = e / N; const int n = (int)(e % N);
        dst[m * ldd + n] += dcol[m * ldc] * w[n];
    }
}

// ---- positional-encoding backward: d(enc)/d(coord) from the saved sin/cos rows -----------------------------
// enc = [v | sin(2^k v) | cos(2^k v)]_k (nerf_helpers.py:322-349): d sin = 2^k cos, d cos = -2^k sin.
template <int D, int L, int INC>
__device__ __forceinline__ void pe_grad(const float *enc, const float *denc, float *dv)
{
    constexpr int D0 = INC ? D : 0;    // include_input: the raw coordinates come first
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float g = INC ? denc[a] : 0.0f;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int si = D0 + 2 * D * k + a, ci = si + D;
            g += (float)(1 << k) * (denc[si] * enc[ci] - denc[ci] * enc[si]);
        }
        dv[a] = g;
    }
}

constexpr int DIN_LD = 16 * (KB_XYZ + KB_AMB);   // row of the gradient wrt [PE(x') blocks | PE(w) blocks]: 96 | 128 | 64
constexpr int DIN_AMB = 16 * KB_XYZ;             // where the PE(w) part starts

// one coordinate's gradient through its encoding (pe_grad, one axis)
template <int D, int L, int INC>
__device__ __forceinline__ float pe_grad_axis(const float *enc, const float *denc, int a)
{
    constexpr int D0 = INC ? D : 0;
    float g = INC ? denc[a] : 0.0f;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int si = D0 + 2 * D * k + a, ci = si + D;
        g += (float)(1 << k) * (denc[si] * enc[ci] - denc[ci] * enc[si]);
    }
    return g;
}

// Per sample: d_in [P x DIN_LD] = gradient wrt [PE(x') blocks | PE(w) blocks] (d_in2, optional: a second contribution, added).  d_xw [P x 4] +=
// dL/dx' through PE63 (the trilinear part was written by grid_backward_kernel), d_w [P x 4] = dL/dw; seam8 (optional): the same as (P,8) rows
// [dx'0 dx'1 dx'2 0 | dw0 dw1 0 0], the form in which the seam gradient leaves the radiance part of a split walk.
// A workgroup takes 64 samples at a time: their gradient rows and saved encodings (contiguous blocks of the planes) come in as coalesced
// 16-byte loads and are parked in LDS; then four threads per sample -- one per coordinate of x', one for w -- walk the octaves.  (Round 4,
// first form: one thread per sample reading its own 384-byte rows, a line per lane and load: 1.5 TB/s of useful bytes, 270 us per step.)
constexpr int EB_SAMPLES = 64, EB_ROW = DIN_LD + 4;      // LDS row stride (floats): 16-byte aligned rows, the four threads of a sample on four banks of their own
__global__ void __launch_bounds__(256) encode_backward_kernel(long P, const float *__restrict__ actbuf, const float *__restrict__ d_in,
                                                              const float *__restrict__ d_in2, float *__restrict__ d_xw, float *__restrict__ d_w,
                                                              float *__restrict__ seam8)
{
    __shared__ __attribute__((aligned(16))) float s_din[EB_SAMPLES * EB_ROW], s_enc[EB_SAMPLES * EB_ROW];
    const int tid = threadIdx.x, sl = tid >> 2, r = tid & 3;
    const long nblk = (P + EB_SAMPLES - 1) / EB_SAMPLES;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long p0 = blk * EB_SAMPLES;
        const int nv = (int)((P - p0 < EB_SAMPLES) ? P - p0 : EB_SAMPLES);
        __syncthreads();      // (the previous block's readers are done)
        for (int i = tid; i < nv * (DIN_LD / 4); i += 256) {
            const int row = i / (DIN_LD / 4), c4 = i - row * (DIN_LD / 4);
            f32x4 v = *reinterpret_cast<const f32x4 *>(d_in + p0 * DIN_LD + 4 * (long)i);
            if (d_in2 != nullptr) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(d_in2 + p0 * DIN_LD + 4 * (long)i);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
            *reinterpret_cast<f32x4 *>(s_din + row * EB_ROW + 4 * c4) = v;
        }
        for (int i = tid; i < nv * (4 * KB_XYZ); i += 256) {
            const int row = i / (4 * KB_XYZ), c4 = i - row * (4 * KB_XYZ);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(actbuf + (long)act::PEX * P + p0 * (16 * KB_XYZ) + 4 * (long)i);
            *reinterpret_cast<f32x4 *>(s_enc + row * EB_ROW + 4 * c4) = v;
        }
#if SAHS_MODEL != 2
        for (int i = tid; i < nv * (4 * KB_AMB); i += 256) {
            const int row = i / (4 * KB_AMB), c4 = i - row * (4 * KB_AMB);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(actbuf + (long)act::PEW * P + p0 * (16 * KB_AMB) + 4 * (long)i);
            *reinterpret_cast<f32x4 *>(s_enc + row * EB_ROW + DIN_AMB + 4 * c4) = v;
        }
#endif
        __syncthreads();
        if (sl < nv) {
            const long p = p0 + sl;
            const float *enc = s_enc + sl * EB_ROW, *din = s_din + sl * EB_ROW;
            if (r < 3) {
                const float t = d_xw[p * 4 + r] + pe_grad_axis<3, L_XYZ, 1>(enc, din, r);
                d_xw[p * 4 + r] = t;
                if (seam8 != nullptr) seam8[p * 8 + r] = t;
            } else {
                float gw[2] = {0.0f, 0.0f};
#if SAHS_MODEL != 2
#pragma unroll
                for (int a = 0; a < AMB_DIM; ++a) gw[a] = pe_grad_axis<AMB_DIM, L_AMB, AMB_INC>(enc + DIN_AMB, din + DIN_AMB, a);
#endif
                *reinterpret_cast<f32x4 *>(d_w + p * 4) = f32x4{gw[0], gw[1], 0.0f, 0.0f};
                if (seam8 != nullptr) {
                    seam8[p * 8 + 3] = d_xw[p * 4 + 3];
                    *reinterpret_cast<f32x4 *>(seam8 + p * 8 + 4) = f32x4{gw[0], gw[1], 0.0f, 0.0f};
                }
            }
        }
    }
}

// Trilinear feature-grid backward (ATen grid_sampler_3d backward, align_corners=True, zeros padding; models.py:346-365).
// Half a wave per sample, lane = channel: the 32 channel gradients of one corner are one 128-byte atomic burst into the
// CHANNEL-LAST accumulator d_grid_cl [voxel][32] (transposed into the channel-first parameter gradient afterwards); the
// coordinate gradient is the 32-lane reduction of dg * grid.  Writes d_xw[p][0:3] = trilinear part, [3] = 0.
__global__ void __launch_bounds__(256) grid_backward_kernel(long P, const float *__restrict__ actbuf, const float *__restrict__ d_gridf,
                                                            const float *__restrict__ grid_cl, float *__restrict__ d_grid_cl,
                                                            float *__restrict__ d_xw)
{
    // Consecutive samples are consecutive depths of one ray and a ray crosses a 1/16-wide cell in ~13 of its 128 samples, so each
    // half-wave walks a run of GCH samples and keeps the eight corner contributions of the CURRENT cell in registers, flushing
    // them (one 128-byte atomic burst per corner) only when the cell changes: ~10x fewer atomics than one flush per sample.
    constexpr int GCH = 16;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const float R1 = (float)(G_RES - 1);
    auto flush = [&](int key, const float *acc) {
        if (key < 0) return;
        const int xi = (key & 255) - 2, yi = ((key >> 8) & 255) - 2, zi = (key >> 16) - 2;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int cx = xi + (n & 1), cy = yi + ((n >> 1) & 1), cz = zi + (n >> 2);
            if (cx >= 0 && cx < G_RES && cy >= 0 && cy < G_RES && cz >= 0 && cz < G_RES)
                atomicAdd(d_grid_cl + (((long)cz * G_RES + cy) * G_RES + cx) * D_GRID + c, acc[n]);
        }
    };
    for (long chunk = wave * 2 + h; chunk * GCH < P; chunk += nwaves * 2) {
        float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int cur = -1;
        for (int k = 0; k < GCH; ++k) {
            const long p = chunk * GCH + k;
            const bool live = p < P;
            const long pc = live ? p : P - 1;
            const float *a = actbuf + (long)act::XW * P + pc * 16;
            const float x = a[0], y = a[1], z = a[2];
            const float ix = ((x + 1.0f) / 2.0f) * R1, iy = ((y + 1.0f) / 2.0f) * R1, iz = ((z + 1.0f) / 2.0f) * R1;
            const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
            const float wx0 = (fx + 1.0f) - ix, wx1 = ix - fx, wy0 = (fy + 1.0f) - iy, wy1 = iy - fy, wz0 = (fz + 1.0f) - iz, wz1 = iz - fz;
            const bool ok = live && fx >= -1.0f && fx <= (float)G_RES && fy >= -1.0f && fy <= (float)G_RES && fz >= -1.0f && fz <= (float)G_RES;
            const int xi = ok ? (int)fx : -2, yi = ok ? (int)fy : -2, zi = ok ? (int)fz : -2;
            const int key = ok ? (((zi + 2) << 16) | ((yi + 2) << 8) | (xi + 2)) : -1;
            if (key != cur) {
                flush(cur, acc);
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[n] = 0.0f;
                cur = key;
            }
            const float g = live ? d_gridf[pc * 32 + c] : 0.0f;
            float gix = 0.0f, giy = 0.0f, giz = 0.0f;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int bx = n & 1, by = (n >> 1) & 1, bz = n >> 2;
                const int cx = xi + bx, cy = yi + by, cz = zi + bz;
                const bool in = ok && cx >= 0 && cx < G_RES && cy >= 0 && cy < G_RES && cz >= 0 && cz < G_RES;
                const float wxb = bx ? wx1 : wx0, wyb = by ? wy1 : wy0, wzb = bz ? wz1 : wz0;
                acc[n] += g * ((wxb * wyb) * wzb);
                if (in) {
                    const float t = g * grid_cl[(((long)cz * G_RES + cy) * G_RES + cx) * D_GRID + c];
                    gix += t * (bx ? 1.0f : -1.0f) * wyb * wzb;
                    giy += t * (by ? 1.0f : -1.0f) * wxb * wzb;
                    giz += t * (bz ? 1.0f : -1.0f) * wxb * wyb;
                }
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                gix += __shfl_xor(gix, off, 64);
                giy += __shfl_xor(giy, off, 64);
                giz += __shfl_xor(giz, off, 64);
            }
            if (live && c == 0) {
                const float sc = R1 / 2.0f;
                *reinterpret_cast<f32x4 *>(d_xw + p * 4) = f32x4{gix * sc, giy * sc, giz * sc, 0.0f};
            }
        }
        flush(cur, acc);
    }
}

// channel-first [32][vox] <-> channel-last [vox][32] (mode 0: dst_cl = src_cf;  mode 1: dst_cf += src_cl, skipping blocks of 32 voxels that
// received no gradient: the rays of a batch cross a few per cent of the grid, and 4 MB of float atomics per walk cost 0.3-1.6 ms)
__global__ void __launch_bounds__(256) grid_transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, int mode)
{
    __shared__ float t[32][33];
    const long vox = (long)G_RES * G_RES * G_RES;
    const long v0 = (long)blockIdx.x * 32;
    const int i = threadIdx.x & 31, j = threadIdx.x >> 5;   // 8 rows per pass
    if (mode == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int ch = j + 8 * r; t[ch][i] = src[(long)ch * vox + v0 + i]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int vv = j + 8 * r; dst[(v0 + vv) * 32 + i] = t[i][vv]; }
    } else {
        int any = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int vv = j + 8 * r; const float v = src[(v0 + vv) * 32 + i]; t[vv][i] = v; any |= (v != 0.0f); }
        if (!__syncthreads_or(any)) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // (both levels' walks add here, possibly at once)
            const int ch = j + 8 * r;
            const float v = t[i][ch];
            if (v != 0.0f) atomicAdd(dst + (long)ch * vox + v0 + i, v);
        }
    }
}

// g3[p][i] = d_xw[p][i] * (1 - dx_i^2)   (x' = x + tanh(.), models.py:304-305)
__global__ void tanh_backward_kernel(long P, const float *__restrict__ actbuf, const float *__restrict__ d_xw, float *__restrict__ g3)
{
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float dx = actbuf[(long)act::DX * P + p * 16 + i];
            g3[p * 4 + i] = d_xw[p * 4 + i] * (1.0f - dx * dx);
        }
        g3[p * 4 + 3] = 0.0f;
    }
}

// per-frame constants: dW[r][col0 + k] += db[r] * c[k];  dc[k] += sum_r W[r][col0 + k] * db[r].  One workgroup per column k.
__global__ void __launch_bounds__(256) const_cols_backward_kernel(int rows, int cols, const float *__restrict__ W, float *__restrict__ dW, long ld,
                                                                  int col0, const float *__restrict__ db, const float *__restrict__ c,
                                                                  float *__restrict__ dc)
{
    const int k = blockIdx.x;
    const float ck = c[k];
    float s = 0.0f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float g = db[r];
        s += W[(long)r * ld + col0 + k] * g;
        atomicAdd(dW + (long)r * ld + col0 + k, g * ck);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dc + k, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void axpy_kernel(int n, const float *__restrict__ x, float *__restrict__ y)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(y + i, x[i]);
}

// ---- the walk's small launches, batched: job lists travel as kernel arguments (no table upload, nothing allocated) ----
// Everything a walk adds into the SHARED gradient buffers (grad_flat, grad_cond) is an atomicAdd: two walks -- the two levels' radiance parts, the
// two deformation parts -- may run at once on two streams (ops.RenderRaysFn.backward).
// (1) 16-byte aligned copies of the weight sub-matrices the data-gradient GEMMs stream by LDS-DMA: all of a walk's copies in ONE launch
//     in front of it (the walk is run once dry to collect them); (2) the per-frame-constant columns and (3) the bias gradients that went
//     through scratch: their results are read by nothing inside the walk, so they are deferred to one launch each at its end.
struct CopyJob { const float *src; float *dst; long lds_, ldd; int K, N; int pack; };
// pack = 1: dst is not a plain copy but the weight sub-matrix W[K x N] (K a multiple of 16) as the data-gradient GEMM's B operand, split and
// in MFMA fragment order: per (16-row K-step, 128-column block) one 8 KB piece [hi: 4 n-tiles x 64 lanes x 8 bf16 | lo: the same], lane
// (r, h) of n-tile t holding W[16 step + 8h + j][128 block + 32 t + r], j = 0..7 -- so the kernel's B tile is a linear 8 KB copy and a
// fragment is one ds_read_b128, with no conversion work in the GEMM (columns past N are zero).  Pieces in (step, block) order.
struct ConstJob { const float *W; float *dW; const float *db, *c; float *dc; long ld; int rows, cols, col0; };
struct AxpyJob { const float *x; float *y; int n; };
constexpr int MAX_COPY_JOBS = 48, MAX_CONST_JOBS = 24, MAX_AXPY_JOBS = 24;
struct CopyBatch { CopyJob j[MAX_COPY_JOBS]; };
struct ConstBatch { ConstJob j[MAX_CONST_JOBS]; };
struct AxpyBatch { AxpyJob j[MAX_AXPY_JOBS]; };

__global__ void __launch_bounds__(256) copy2d_batch_kernel(CopyBatch b, int phase)      // phase 0: the plain copies, 1: the pack jobs
{                                                                                        // (a pack job may read what a plain copy wrote)
    const CopyJob &j = b.j[blockIdx.y];
    if ((j.pack != 0) != (phase != 0)) return;
    if (j.pack) {
        const int nblk = (j.N + GT - 1) / GT;
        const long total = (long)(j.K / 2) * nblk * GT;        // one thread per (row pair, padded column)
        uint32_t *dst = reinterpret_cast<uint32_t *>(j.dst);
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const int n = (int)(e % (nblk * GT));
            const int k = 2 * (int)(e / (nblk * GT));
            const float x0 = n < j.N ? j.src[(long)k * j.lds_ + n] : 0.0f, x1 = n < j.N ? j.src[(long)(k + 1) * j.lds_ + n] : 0.0f;
            const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
            const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{x0 - __builtin_bit_cast(float, hi << 16),
                                                                                                 x1 - __builtin_bit_cast(float, hi & 0xffff0000u)}, bf16x2_t));
            const int step = k >> 4, kk = k & 15, h = kk >> 3, jp = (kk & 7) >> 1, cb = n >> 7, t = (n & 127) >> 5, r = n & 31;
            uint32_t *piece = dst + ((long)step * nblk + cb) * (GK * GT);       // 2048 dwords = 8 KB
            const int at = (t * 64 + h * 32 + r) * 4 + jp;                      // dword within the hi (or lo) half
            piece[at] = hi;
            piece[GK * GT / 2 + at] = lo;
        }
        return;
    }
    const long total = (long)j.K * j.N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long m = e / j.N; const int n = (int)(e % j.N);
        j.dst[m * j.ldd + n] = j.src[m * j.lds_ + n];
    }
}

// one workgroup per job: thread (g, k) walks rows g, g + ngroups, ... of column k -- the reads of W and the atomics into dW are contiguous
// along k (a block per column made them 256 strided accesses each), dc[k] is reduced over the row groups through LDS
__global__ void __launch_bounds__(256) const_cols_batch_kernel(ConstBatch b)
{
    const ConstJob &j = b.j[blockIdx.x];
    __shared__ float part[256];
    const int cols = j.cols, ngroups = 256 / cols, g = threadIdx.x / cols, k = threadIdx.x - g * cols;
    float s = 0.0f;
    if (g < ngroups) {
        const float ck = j.c[k];
        for (int r = g; r < j.rows; r += ngroups) {
            const float gr = j.db[r];
            s += j.W[(long)r * j.ld + j.col0 + k] * gr;
            atomicAdd(j.dW + (long)r * j.ld + j.col0 + k, gr * ck);
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cols) {
        float t = 0.0f;
        for (int q = 0; q < ngroups; ++q) t += part[q * cols + threadIdx.x];
        atomicAdd(j.dc + threadIdx.x, t);      // several jobs share a dc (d driving, d pose)
    }
}

__global__ void __launch_bounds__(256) axpy_batch_kernel(AxpyBatch b)
{
    const AxpyJob &j = b.j[blockIdx.y];
    for (int i = threadIdx.x; i < j.n; i += blockDim.x) atomicAdd(j.y + i, j.x[i]);
}

}  // namespace SAHS_NS

using namespace SAHS_NS;

// Precision of the backward GEMMs of this model build: 0 = f32 MFMA (exact products), 3 = bf16 pipe with split operands (default;
// SAHS_BWD_GEMM=f32 in the environment selects 0 at first use).  set < 0 queries.  Process-wide (one atomic), set through
// sahs_backward_gemm_precision() of the C ABI.
extern "C" int SAHS_SYM(sahs_bwd_gemm_precision_state)(int set)
{
    static std::atomic<int> state{-1};
    int cur = state.load(std::memory_order_relaxed);
    if (cur < 0) {
        const char *e = getenv("SAHS_BWD_GEMM");
        cur = (e != nullptr && e[0] == 'f') ? 0 : 3;
        state.store(cur, std::memory_order_relaxed);
    }
    if (set >= 0) { state.store(set ? 3 : 0, std::memory_order_relaxed); cur = set ? 3 : 0; }
    return cur;
}

namespace {

struct Bwd {
    hipStream_t st;
    long P;
    const float *zero;     // 16-byte zero page (DMA source for out-of-range tile rows/columns)
    float *wal;            // scratch for 16-byte aligned copies of weight sub-matrices (the flat parameter buffer is not aligned)
    long wal_cap = 0;
    long walo = 0;
    int err = 0;
    unsigned char *sign_bits = nullptr;      // P x 32 bytes: the bit matrix of the activation the last tn() staged as its X operand ...
    const float *bits_of = nullptr;          // ... which is this matrix, N = bits_n columns (null: none)
    int bits_n = 0;
    bool dry = false;      // the collecting pass: nothing is launched, nn() records the aligned copies it will need
    CopyBatch copies; int ncopy = 0;
    ConstBatch consts; int nconst = 0, const_maxcols = 0;
    AxpyBatch axpys; int naxpy = 0;
    void check() { if (!err) err = (int)hipGetLastError(); }
    void flush_copies()
    {
        // more jobs than a batch holds would leave the heads' zero-padded weight copies unmade (head_copy only records): fail loudly
        if (ncopy > MAX_COPY_JOBS) { if (!err) err = (int)hipErrorOutOfMemory; return; }
        if (ncopy > 0) {
            copy2d_batch_kernel<<<dim3(16, ncopy), 256, 0, st>>>(copies, 0); check();
            copy2d_batch_kernel<<<dim3(16, ncopy), 256, 0, st>>>(copies, 1); check();
        }
    }
    void flush_deferred()
    {
        if (nconst > 0) { const_cols_batch_kernel<<<nconst, 256, 0, st>>>(consts); check(); }      // (cols <= 256: D_DRV 76, D_POSE 36)
        if (naxpy > 0) { axpy_batch_kernel<<<dim3(1, naxpy), 256, 0, st>>>(axpys); check(); }
        nconst = naxpy = 0;
    }
    static bool x3() { return SAHS_SYM(sahs_bwd_gemm_precision_state)(-1) != 0; }
    static bool al(const void *p, long ld)
    {
        static const bool nodma = getenv("SAHS_BWD_NODMA") != nullptr;     // diagnostic: route every GEMM to the register-staged kernel
        return !nodma && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0;
    }
    // dX[P x N] (mode) = dY[P x K] * W[K x N] (* mask)
    void nn(const float *dY, long ldy, int K, const float *W, long ldw, int N, float *dX, long ldx, int mode, const float *mask = nullptr,
            long ldm = 0, float slope = 0.0f)
    {
        const int nbn = (N + GT - 1) / GT;
        const long nbm = (P + GT - 1) / GT;
        dim3 g((unsigned)(((nbm + 7) / 8) * 8 * nbn), 1, 1);
        if (K % GK == 0 && al(dY, ldy)) {
            if (x3()) {      // the B operand pre-split into bf16 hi / lo in MFMA fragment order (CopyJob pack), one batched launch per walk
                const long words = (long)K * nbn * GT;
                if (walo + words > wal_cap) { if (!err) err = (int)hipErrorOutOfMemory; return; }
                float *dst = wal + walo;
                walo += words;
                if (dry) {
                    if (ncopy < MAX_COPY_JOBS) copies.j[ncopy] = CopyJob{W, dst, ldw, 0, K, N, 1};
                    ++ncopy;
                    return;
                }
                if (ncopy > MAX_COPY_JOBS) {       // more jobs than a batch holds: pack one by one
                    CopyBatch one;
                    one.j[0] = CopyJob{W, dst, ldw, 0, K, N, 1};
                    copy2d_batch_kernel<<<dim3(16, 1), 256, 0, st>>>(one, 1);
                    check();
                }
                W = dst; ldw = 0;
            } else if (!al(W, ldw)) {
                const long ldb = (N + 3) / 4 * 4;
                if (walo + (long)K * ldb > wal_cap) { if (!err) err = (int)hipErrorOutOfMemory; return; }   // scratch sized for one level's weights
                float *dst = wal + walo;
                walo += (long)K * ldb;
                if (dry) {
                    if (ncopy < MAX_COPY_JOBS) copies.j[ncopy] = CopyJob{W, dst, ldw, ldb, K, N, 0};
                    ++ncopy;       // (more than MAX_COPY_JOBS: the real pass copies one by one, as before)
                    return;
                }
                if (ncopy > MAX_COPY_JOBS) {
                    const long tot = (long)K * N;
                    copy2d_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(K, N, W, ldw, dst, ldb, 0);
                    check();
                }
                W = dst; ldw = ldb;
            }
            if (dry) return;
            // the layer's weight-gradient GEMM has just staged this very mask matrix and left its sign bits (tn)
            unsigned char *mb = (mask != nullptr && mask == bits_of && N == bits_n && N % 8 == 0) ? sign_bits : nullptr;
            if (x3()) gemm_dma_kernel<false, true><<<g, 256, 0, st>>>((int)P, N, K, dY, ldy, W, ldw, dX, ldx, mode, mask, ldm, slope, K, nullptr, nbn, zero, mb);
            else gemm_dma_kernel<false, false><<<g, 256, 0, st>>>((int)P, N, K, dY, ldy, W, ldw, dX, ldx, mode, mask, ldm, slope, K, nullptr, nbn, zero, mb);
        } else {
            if (dry) return;
            gemm_f32_kernel<false><<<g, 256, 0, st>>>((int)P, N, K, dY, ldy, W, ldw, dX, ldx, mode, mask, ldm, slope, K, nullptr, nbn);
        }
        check();
    }
    // dW[M x N] += dY[P x M]^T * X[P x N];  db != null: db[M] += column sums of dY (fused: the dY tiles are staged anyway).
    // The sample dimension is cut into slabs of >= 512 samples, ~2 workgroups per CU per launch: a workgroup's fixed cost (ring fill,
    // 64 atomics per lane) wants many K-steps per slab -- measured on the 2048-ray step: 1536 workgroups / 256-sample slabs 35.6 ms,
    // 512 / 512 33.0 ms, 256 / 1024 34.5 ms; writing partial tiles and reducing them in a second pass instead of the atomics
    // changed nothing at any setting (the atomics are not the cost).
    void tn(const float *dY, long ldy, int M, const float *X, long ldx, int N, float *dW, long ldw, float *db = nullptr)
    {
        if (dry) return;
        const int tiles = ((N + GT - 1) / GT) * ((M + GT - 1) / GT);
        static const long wg_target = getenv("SAHS_BWD_TN_WGS") ? atol(getenv("SAHS_BWD_TN_WGS")) : 512;      // (tuning aid)
        long kslab = (P * tiles / wg_target + 15) / 16 * 16;
        static const long min_slab = getenv("SAHS_BWD_TN_MINSLAB") ? atol(getenv("SAHS_BWD_TN_MINSLAB")) : 512;      // (tuning aid)
        kslab = kslab < min_slab ? min_slab : (kslab > 8192 ? 8192 : kslab);
        dim3 g((N + GT - 1) / GT, (M + GT - 1) / GT, (unsigned)((P + kslab - 1) / kslab));
        if (al(dY, ldy) && al(X, ldx)) {
            const int nx = (N + GT - 1) / GT, slabs = (int)((P + kslab - 1) / kslab);
            const dim3 g1((unsigned)((slabs + 7) / 8 * 8 * tiles), 1, 1);        // XCD-aware 1-D grid (kernel: nbn < 0)
            unsigned char *mb = (sign_bits != nullptr && N % 8 == 0 && N <= 256) ? sign_bits : nullptr;
            static const bool dbg_noatomic = getenv("SAHS_BWD_DBG_NOATOMIC") != nullptr;      // timing experiment: results wrong
            if (dbg_noatomic) ldw = 0;
            static const bool split_once = getenv("SAHS_BWD_TN_INLINE_SPLIT") == nullptr;      // (A/B aid: the per-wave split of gemm_dma_kernel<true, true>)
            if (x3() && split_once) {
                static sahs_once::Flags attr_set;       // the large-LDS attribute is per device
                const hipError_t ae = sahs_once::per_device(attr_set, [&]() {
                    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
                });
                if (ae != hipSuccess) { if (!err) err = (int)ae; return; }
                gemm_tn_split_kernel<<<g1, TN_THREADS, TN_LDS_BYTES, st>>>(M, N, (int)P, dY, ldy, X, ldx, dW, ldw, (int)kslab, db, nx, zero, mb);
            } else if (x3()) gemm_dma_kernel<true, true><<<g1, 256, 0, st>>>(M, N, (int)P, dY, ldy, X, ldx, dW, ldw, 2, nullptr, 0, 0.0f, (int)kslab, db, -nx, zero, mb);
            else gemm_dma_kernel<true, false><<<g1, 256, 0, st>>>(M, N, (int)P, dY, ldy, X, ldx, dW, ldw, 2, nullptr, 0, 0.0f, (int)kslab, db, -nx, zero, mb);
            bits_of = mb ? X : nullptr;
            bits_n = N;
        } else
        {
            gemm_f32_kernel<true><<<g, 256, 0, st>>>(M, N, (int)P, dY, ldy, X, ldx, dW, ldw, 2, nullptr, 0, 0.0f, (int)kslab, db, 0);
            bits_of = nullptr;
        }
        check();
    }
    void copy(const float *src, long lds_, int N, float *dst, long ldd, int mode)
    {
        if (dry) return;
        copy2d_kernel<<<2048, 256, 0, st>>>(P, N, src, lds_, dst, ldd, mode);
        check();
    }
};

}  // namespace

// Words of workspace per call: gA, gB, dfeat (P x 256 each), din (P x 96), dgridf (P x 32), dxw, dw, g3 (P x 4 each), db scratch,
// channel-last copies of the feature grid and of its gradient accumulator
constexpr int DB_SCRATCH = 8192;   // per-call bias-gradient scratch (all layers of one level: ~5.3 K floats)
constexpr long WAL_FLOATS = 2L << 20;   // aligned weight sub-matrix copies of one level (< 1.8 M floats)
// the three output heads as 16-row matrices over the whole d_raw row [drgb3 | dseg12 | dsigma]: zero-padded weights W16 and their gradients
constexpr int HEAD_W_SEG = 0, HEAD_W_RGB = 16 * 128, HEAD_W_ALPHA = 2 * 16 * 128, HEAD_G_SEG = HEAD_W_ALPHA + 16 * 256, HEAD_G_RGB = HEAD_G_SEG + 16 * 128,
              HEAD_G_ALPHA = HEAD_G_RGB + 16 * 128, HEAD_DB = HEAD_G_ALPHA + 16 * 256;
constexpr long HEAD_FLOATS = HEAD_DB + 64;
extern "C" long SAHS_SYM(sahs_field_backward_ws_words)(long P) { return P * (256L * 3 + DIN_LD + 32 + 12) + DB_SCRATCH + 2 * GRID_FLOATS + WAL_FLOATS + P * 8 + HEAD_FLOATS; }

// grad_cond: [0:76] d_driving, [80:116] d_pose36 (accumulated).  grad_flat: accumulated.  d_raw: (P,16).
// part (bit 1: deformation nets, bit 2: radiance nets; 0 = 3 = everything) cuts the walk at its seam, the gradient w.r.t. the deformed
// point and the ambient coordinate, (P,8) rows [dx'0 dx'1 dx'2 . dw0 dw1 . .]: the radiance part alone leaves it in xwg_out, the
// deformation part alone starts from xwg_in, and the whole walk adds xwg_in (if given) at the seam -- that is how the fine pass's
// gradient reaches the coarse samples' deformation when the forward evaluated the deformation nets once per depth (field_f32.hip MODE).
extern "C" int SAHS_SYM(sahs_field_backward_split_launch)(const float *flat, const float *frame, int level, int part, long P, const float *actbuf,
                                                const float *d_raw, const float *xwg_in, float *xwg_out, float *grad_flat, float *grad_cond,
                                                float *ws, hipStream_t stream);
extern "C" int SAHS_SYM(sahs_field_backward_launch)(const float *flat, const float *frame, int level, long P, const float *actbuf, const float *d_raw,
                                          float *grad_flat, float *grad_cond, float *ws, hipStream_t stream)
{
    return SAHS_SYM(sahs_field_backward_split_launch)(flat, frame, level, 3, P, actbuf, d_raw, nullptr, nullptr, grad_flat, grad_cond, ws, stream);
}
extern "C" int SAHS_SYM(sahs_field_backward_split_launch)(const float *flat, const float *frame, int level, int part, long P, const float *actbuf,
                                                const float *d_raw, const float *xwg_in, float *xwg_out, float *grad_flat, float *grad_cond,
                                                float *ws, hipStream_t stream)
{
    if (P <= 0) return 0;
    if (part == 0) part = 3;
    const bool do_rad = (part & 2) != 0, do_def = (part & 1) != 0;
#if SAHS_MODEL == 2
    if (part != 3) return -3;      // no deformation nets: nothing to split
#endif
    if (P > 4000000L) return -3;   // gridDim.y of the P x N GEMMs; callers chunk larger batches
    Bwd b{stream, P};
    b.zero = ws + P * (256L * 3 + DIN_LD + 32 + 12) + DB_SCRATCH - 64;   // tail of the (zeroed) bias-gradient scratch, never written
    b.wal = ws + P * (256L * 3 + DIN_LD + 32 + 12) + DB_SCRATCH + 2 * GRID_FLOATS;
    b.wal_cap = WAL_FLOATS;
    float *heads = b.wal + WAL_FLOATS + P * 8;
    if (hipMemsetAsync(heads, 0, sizeof(float) * HEAD_FLOATS, stream) != hipSuccess) return (int)hipGetLastError();
    b.sign_bits = getenv("SAHS_BWD_NOBITS") ? nullptr : reinterpret_cast<unsigned char *>(b.wal + WAL_FLOATS);     // P x 32 bytes (<= 256 columns)
    const FlatOffsets &F = kFlat;
    const FlatOffsets::Lvl &Lv = F.lvl[level];
    float *gA = ws, *gB = gA + P * 256, *dfeat = gB + P * 256, *din = dfeat + P * 256, *dgridf = din + P * DIN_LD, *dxw = dgridf + P * 32,
          *dw = dxw + P * 4, *g3 = dw + P * 4, *db = g3 + P * 4, *grid_cl = db + DB_SCRATCH, *dgrid_cl = grid_cl + GRID_FLOATS;
    // saved activations: one dense [P x width] array per layer, the array at column c of the act:: table starting at c * P
    if (hipMemsetAsync(db, 0, sizeof(float) * DB_SCRATCH, stream) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(dgrid_cl, 0, sizeof(float) * GRID_FLOATS, stream) != hipSuccess) return (int)hipGetLastError();
    const float *drv = frame + FRAME_DRV_OFF, *p36 = frame + FRAME_POSE_OFF;
    float *d_drv = grad_cond + 0, *d_p36 = grad_cond + 80;
    const float *trc = TRUNK_SEES_POSE ? p36 : drv;          // the per-frame constant the trunk sees, and its gradient slot
    float *d_trc = TRUNK_SEES_POSE ? d_p36 : d_drv;
    auto W = [&](long off) { return flat + off; };
    auto G = [&](long off) { return grad_flat + off; };
    int dbo = 0;   // running offset into db scratch
    // The walk runs twice: once dry -- nothing is launched, the data-gradient GEMMs record which weight sub-matrices they need as
    // 16-byte aligned copies -- then all of those copies in one launch, then for real.  Both passes take the same branches, so the
    // scratch offsets (walo, dbo) they hand out are the same.
    auto walk = [&]() -> int {
    dbo = 0;
    b.walo = 0;
    // bias-gradient slots of one level: BIAS_FLOATS-ish; the last 64 floats of the scratch are the DMA zero page
    static_assert(BIAS_FLOATS + 16 * 40 <= DB_SCRATCH - 64, "bias-gradient scratch too small for this model");
    // bias gradient of a layer: straight into the flat gradient (boff >= 0), or -- for the six layers whose folded per-frame
    // constants need this call's db on its own -- into a scratch slot that add_bias then adds
    auto newdb = [&](int n, long boff = -1) { if (boff >= 0) return G(boff); float *p = db + dbo; dbo += (n + 3) / 4 * 4; return p; };
    auto add_bias = [&](float *dbl, long boff, int n) {      // deferred to the end of the walk, batched (nothing in the walk reads G(boff))
        if (dbl == G(boff) || b.dry) return;
        if (b.naxpy < MAX_AXPY_JOBS) { b.axpys.j[b.naxpy++] = AxpyJob{dbl, G(boff), n}; return; }
        axpy_kernel<<<1, 256, 0, stream>>>(n, dbl, G(boff)); b.check();
    };
    auto consts = [&](long woff, long ld, int rows, int col0, int cols, const float *dbl, const float *c, float *dc) {
        if (b.dry) return;
        if (b.nconst < MAX_CONST_JOBS) {                     // deferred as well: dW's constant columns and dc are read by nothing in the walk
            b.consts.j[b.nconst++] = ConstJob{W(woff), G(woff), dbl, c, dc, ld, rows, cols, col0};
            b.const_maxcols = cols > b.const_maxcols ? cols : b.const_maxcols;
            return;
        }
        const_cols_backward_kernel<<<cols, 256, 0, stream>>>(rows, cols, W(woff), G(woff), ld, col0, dbl, c, dc);
        b.check();
    };
    // generic plain layer: given dY (pre-activation grads of this layer, P x out) and its input X (P x in, stride ldx):
    //   dW += dY^T X, db += colsum, and (if dX) dX = dY W * mask(prevact)
    auto layer_params = [&](const float *dY, long ldy, int out, long woff, long boff, long ldw, int col0, const float *X, long ldx, int in,
                            float *dbl) {
        b.tn(dY, ldy, out, X, ldx, in, G(woff) + col0, ldw);
        (void)boff; (void)dbl;
    };

    // rows of a head's weights into its zero-padded 16-row copy (collected in the dry pass, one batched launch with the aligned weight
    // copies); rows of a head's gradient scratch added to the flat gradient (deferred to the end of the walk, batched)
    auto head_copy = [&](const float *src, float *dst, int rows, int cols) {
        if (!b.dry) return;
        if (b.ncopy < MAX_COPY_JOBS) b.copies.j[b.ncopy] = CopyJob{src, dst, cols, cols, rows, cols, 0};
        else b.err = (int)hipErrorOutOfMemory;
        ++b.ncopy;
    };
    auto head_add = [&](const float *src, float *dst, int n) {
        if (b.dry) return;
        if (b.naxpy < MAX_AXPY_JOBS) { b.axpys.j[b.naxpy++] = AxpyJob{src, dst, n}; return; }
        axpy_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, src, dst); b.check();
    };
    const float *A = actbuf;
    if (do_rad) {
    // ================= seg branch: seg = fc_seg(s3), s_i = lrelu(layers_seg[i](.)) (modules.py:289-294) =================
    {
        // the output heads read the whole 16-float d_raw row ([drgb3 | dseg12 | dsigma], 16-byte aligned, K = 16 = one LDS-DMA K-step)
        // against 16-row zero-padded copies of their weights, instead of its unaligned 3/12/1-column slices through the register-staged
        // GEMM: dW16 = d_raw^T X lands in scratch, rows 3..14 of it are fc_seg's gradient; its column sums are the three bias gradients
        static_assert(BR_H == 128 && TR_H == 256 && N_SEG == 12, "head scratch layout");
        head_copy(W(Lv.segout_w), heads + HEAD_W_SEG + 3 * BR_H, N_SEG, BR_H);
        head_copy(W(Lv.rgb_w), heads + HEAD_W_RGB, 3, BR_H);
        head_copy(W(Lv.alpha_w), heads + HEAD_W_ALPHA + 15 * TR_H, 1, TR_H);
        b.tn(d_raw, 16, 16, A + (long)(act::S + 384) * P, BR_H, BR_H, heads + HEAD_G_SEG, BR_H, heads + HEAD_DB);
        head_add(heads + HEAD_G_SEG + 3 * BR_H, G(Lv.segout_w), N_SEG * BR_H);
        head_add(heads + HEAD_DB + 3, G(Lv.segout_b), N_SEG);
        head_add(heads + HEAD_DB, G(Lv.rgb_b), 3);
        head_add(heads + HEAD_DB + 15, G(Lv.alpha_b), 1);
        b.nn(d_raw, 16, 16, heads + HEAD_W_SEG, BR_H, BR_H, gA, BR_H, 0, A + (long)(act::S + 384) * P, BR_H, 0.01f);
        float *cur = gA, *nxt = gB;
        for (int i = 3; i >= 1; --i) {   // layers_seg[i]: s_{i-1} (128) -> s_i
            float *d = newdb(BR_H, Lv.seg_b[i]);
            b.tn(cur, BR_H, BR_H, A + (long)(act::S + 128 * (i - 1)) * P, BR_H, BR_H, G(Lv.seg_w[i]), BR_H, d);
            add_bias(d, Lv.seg_b[i], BR_H);
            b.nn(cur, BR_H, BR_H, W(Lv.seg_w[i]), BR_H, BR_H, nxt, BR_H, 0, A + (long)(act::S + 128 * (i - 1)) * P, BR_H, 0.01f);
            float *t = cur; cur = nxt; nxt = t;
        }
        float *d = newdb(BR_H, Lv.seg_b[0]);   // layers_seg[0]: feat (256) -> s0
        b.tn(cur, BR_H, BR_H, A + (long)(act::FEAT) * P, TR_H, TR_H, G(Lv.seg_w[0]), TR_H, d);
        add_bias(d, Lv.seg_b[0], BR_H);
        b.nn(cur, BR_H, BR_H, W(Lv.seg_w[0]), TR_H, TR_H, dfeat, 256, 0);       // feat has no activation
    }
    // ================= colour branch (modules.py:276-287) =================
    {
        b.tn(d_raw, 16, 16, A + (long)(act::C + 384) * P, BR_H, BR_H, heads + HEAD_G_RGB, BR_H);
        head_add(heads + HEAD_G_RGB, G(Lv.rgb_w), 3 * BR_H);
        b.nn(d_raw, 16, 16, heads + HEAD_W_RGB, BR_H, BR_H, gA, BR_H, 0, A + (long)(act::C + 384) * P, BR_H, 0.01f);
        float *cur = gA, *nxt = gB;
        for (int i = 3; i >= 1; --i) {
            float *d = newdb(BR_H, Lv.dir_b[i]);
            b.tn(cur, BR_H, BR_H, A + (long)(act::C + 128 * (i - 1)) * P, BR_H, BR_H, G(Lv.dir_w[i]), BR_H, d);
            add_bias(d, Lv.dir_b[i], BR_H);
            b.nn(cur, BR_H, BR_H, W(Lv.dir_w[i]), BR_H, BR_H, nxt, BR_H, 0, A + (long)(act::C + 128 * (i - 1)) * P, BR_H, 0.01f);
            float *t = cur; cur = nxt; nxt = t;
        }
        // layers_dir[0]: [feat256 | dirPE27 | grid32] -> c0
        float *d = newdb(BR_H, Lv.dir_b[0]);
        b.tn(cur, BR_H, BR_H, A + (long)(act::FEAT) * P, TR_H, TR_H, G(Lv.dir_w[0]), D_DIR_IN, d);
        b.tn(cur, BR_H, BR_H, A + (long)(act::DIR) * P, 32, D_DIR, G(Lv.dir_w[0]) + TR_H, D_DIR_IN);
        b.tn(cur, BR_H, BR_H, A + (long)(act::GRID) * P, 32, D_GRID, G(Lv.dir_w[0]) + TR_H + D_DIR, D_DIR_IN);
        add_bias(d, Lv.dir_b[0], BR_H);
        b.nn(cur, BR_H, BR_H, W(Lv.dir_w[0]), D_DIR_IN, TR_H, dfeat, 256, 1);
        b.nn(cur, BR_H, BR_H, W(Lv.dir_w[0]) + TR_H + D_DIR, D_DIR_IN, D_GRID, dgridf, 32, 0);
    }
    // ================= sigma = fc_alpha(feat) (modules.py:275) =================
    {
        b.tn(d_raw, 16, 16, A + (long)(act::FEAT) * P, TR_H, TR_H, heads + HEAD_G_ALPHA, TR_H);
        head_add(heads + HEAD_G_ALPHA + 15 * TR_H, G(Lv.alpha_w), TR_H);
        b.nn(d_raw, 16, 16, heads + HEAD_W_ALPHA, TR_H, TR_H, dfeat, 256, 1);          // d feat += d sigma (x) w_alpha
    }
    // ================= trunk (modules.py:267-274) =================
    {
        // feat = fc_feat(t7)
        float *d = newdb(TR_H, Lv.feat_b);
        b.tn(dfeat, 256, TR_H, A + (long)(act::T + (TR_LAYERS - 1) * 256) * P, TR_H, TR_H, G(Lv.feat_w), TR_H, d);
        add_bias(d, Lv.feat_b, TR_H);
        b.nn(dfeat, 256, TR_H, W(Lv.feat_w), TR_H, TR_H, gA, 256, 0, A + (long)(act::T + (TR_LAYERS - 1) * 256) * P, TR_H, 0.01f);
        float *cur = gA, *nxt = gB;
        for (int i = TR_LAYERS - 1; i >= 1; --i) {   // layers_xyz[i]: input t_{i-1} (and, for i == 3, [PE(x') | PE(w) | pose36])
            float *dl = (i == 3) ? newdb(TR_H) : newdb(TR_H, Lv.xyz_b[i]);
            const long ldw = (i == 3) ? TR_H + D_TR_IN : TR_H;
            b.tn(cur, 256, TR_H, A + (long)(act::T + (i - 1) * 256) * P, TR_H, TR_H, G(Lv.xyz_w[i]), ldw, dl);
            add_bias(dl, Lv.xyz_b[i], TR_H);
            if (i == 3) {
                b.tn(cur, 256, TR_H, A + (long)(act::PEX) * P, 16 * KB_XYZ, D_XYZ, G(Lv.xyz_w[3]) + TR_H, ldw);
                if (D_AMB > 0) b.tn(cur, 256, TR_H, A + (long)(act::PEW) * P, 16 * KB_AMB, D_AMB, G(Lv.xyz_w[3]) + TR_H + D_XYZ, ldw);
                consts(Lv.xyz_w[3], ldw, TR_H, TR_H + D_XYZ + D_AMB, D_TR_CONST, dl, trc, d_trc);
                b.nn(cur, 256, TR_H, W(Lv.xyz_w[3]) + TR_H, ldw, D_XYZ, din, DIN_LD, 0);        // first writer of din stores,
                if (D_AMB > 0) b.nn(cur, 256, TR_H, W(Lv.xyz_w[3]) + TR_H + D_XYZ, ldw, D_AMB, din + DIN_AMB, DIN_LD, 0);   // layers_xyz[0] below accumulates
            }
            b.nn(cur, 256, TR_H, W(Lv.xyz_w[i]), ldw, TR_H, nxt, 256, 0, A + (long)(act::T + (i - 1) * 256) * P, TR_H, 0.01f);
            float *t = cur; cur = nxt; nxt = t;
        }
        // layers_xyz[0]: [PE63(x') | PE18(w) | pose36] -> t0
        float *dl = newdb(TR_H);
        b.tn(cur, 256, TR_H, A + (long)(act::PEX) * P, 16 * KB_XYZ, D_XYZ, G(Lv.xyz_w[0]), D_TR_IN, dl);
        if (D_AMB > 0) b.tn(cur, 256, TR_H, A + (long)(act::PEW) * P, 16 * KB_AMB, D_AMB, G(Lv.xyz_w[0]) + D_XYZ, D_TR_IN);
        add_bias(dl, Lv.xyz_b[0], TR_H);
        consts(Lv.xyz_w[0], D_TR_IN, TR_H, D_XYZ + D_AMB, D_TR_CONST, dl, trc, d_trc);
        b.nn(cur, 256, TR_H, W(Lv.xyz_w[0]), D_TR_IN, D_XYZ, din, DIN_LD, 1);
        if (D_AMB > 0) b.nn(cur, 256, TR_H, W(Lv.xyz_w[0]) + D_XYZ, D_TR_IN, D_AMB, din + DIN_AMB, DIN_LD, 1);
    }
    // ================= encodings + feature grid -> d x', d w =================
    {
        const int tb = (int)(GRID_FLOATS / 32 / 32);   // 32 voxels x 32 channels per block
        if (!b.dry) {
        grid_transpose_kernel<<<tb, 256, 0, stream>>>(W(F.grid), grid_cl, 0); b.check();
        const long gb = (P + 127) / 128;                // 4 waves x 2 runs of 16 samples per block pass
        grid_backward_kernel<<<(unsigned)(gb < 8192 ? gb : 8192), 256, 0, stream>>>(P, actbuf, dgridf, grid_cl, dgrid_cl, dxw); b.check();
        grid_transpose_kernel<<<tb, 256, 0, stream>>>(dgrid_cl, G(F.grid), 1); b.check();
        encode_backward_kernel<<<2048, 256, 0, stream>>>(P, actbuf, din, nullptr, dxw, dw, nullptr); b.check();
        }
    }
    }   // do_rad
    // ---- the seam: d x' (P,4) and d w (P,4) ----
    if (do_rad && !do_def && xwg_out != nullptr) { b.copy(dxw, 4, 4, xwg_out, 8, 0); b.copy(dw, 4, 4, xwg_out + 4, 8, 0); }
    if (xwg_in != nullptr && do_def) { b.copy(xwg_in, 8, 4, dxw, 4, do_rad ? 1 : 0); b.copy(xwg_in + 4, 8, 4, dw, 4, do_rad ? 1 : 0); }
    if (!do_def) { if (!b.dry) b.flush_deferred(); if (!b.err && dbo > DB_SCRATCH - 64) b.err = (int)hipErrorOutOfMemory; return b.err; }
#if SAHS_MODEL != 2
    // ================= hyper sheet (modules.py:444-462): w = fc_ambient(g5) =================
    {
        float *dbl = newdb(4, F.hyp_fb);
        b.tn(dw, 4, AMB_DIM, A + (long)(act::HH + 5 * 64) * P, HYP_H, HYP_H, G(F.hyp_fw), HYP_H, dbl);
        add_bias(dbl, F.hyp_fb, AMB_DIM);
        b.nn(dw, 4, AMB_DIM, W(F.hyp_fw), HYP_H, HYP_H, gA, HYP_H, 0, A + (long)(act::HH + 5 * 64) * P, HYP_H, 0.0f);
        float *cur = gA, *nxt = gB;
        for (int i = 5; i >= 1; --i) {
            float *dl = (i == 4) ? newdb(HYP_H) : newdb(HYP_H, F.hyp_b[i]);
            const long ldw = (i == 4) ? HYP_H + D_DEF_IN : HYP_H;
            b.tn(cur, HYP_H, HYP_H, A + (long)(act::HH + (i - 1) * 64) * P, HYP_H, HYP_H, G(F.hyp_w[i]), ldw, dl);
            add_bias(dl, F.hyp_b[i], HYP_H);
            if (i == 4) {
                b.tn(cur, HYP_H, HYP_H, A + (long)(act::E) * P, 16 * KB_XYZ, D_XYZ, G(F.hyp_w[4]) + HYP_H, ldw);
                consts(F.hyp_w[4], ldw, HYP_H, HYP_H + D_XYZ, D_DRV, dl, drv, d_drv);
                consts(F.hyp_w[4], ldw, HYP_H, HYP_H + D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
            }
            b.nn(cur, HYP_H, HYP_H, W(F.hyp_w[i]), ldw, HYP_H, nxt, HYP_H, 0, A + (long)(act::HH + (i - 1) * 64) * P, HYP_H, 0.0f);
            float *t = cur; cur = nxt; nxt = t;
        }
        float *dl = newdb(HYP_H);
        b.tn(cur, HYP_H, HYP_H, A + (long)(act::E) * P, 16 * KB_XYZ, D_XYZ, G(F.hyp_w[0]), D_DEF_IN, dl);
        add_bias(dl, F.hyp_b[0], HYP_H);
        consts(F.hyp_w[0], D_DEF_IN, HYP_H, D_XYZ, D_DRV, dl, drv, d_drv);
        consts(F.hyp_w[0], D_DEF_IN, HYP_H, D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
    }
    // ================= warp field (modules.py:371-390): x' = x + tanh(fc_final(h5)) =================
    {
        if (!b.dry) { tanh_backward_kernel<<<2048, 256, 0, stream>>>(P, actbuf, dxw, g3); b.check(); }
        float *dbl = newdb(4, F.warp_fb);
        b.tn(g3, 4, 3, A + (long)(act::WH + 5 * 128) * P, WARP_H, WARP_H, G(F.warp_fw), WARP_H, dbl);
        add_bias(dbl, F.warp_fb, 3);
        b.nn(g3, 4, 3, W(F.warp_fw), WARP_H, WARP_H, gA, WARP_H, 0, A + (long)(act::WH + 5 * 128) * P, WARP_H, 0.0f);
        float *cur = gA, *nxt = gB;
        for (int i = 5; i >= 1; --i) {
            float *dl = (i == 4) ? newdb(WARP_H) : newdb(WARP_H, F.warp_b[i]);
            const long ldw = (i == 4) ? WARP_H + D_DEF_IN : WARP_H;
            b.tn(cur, WARP_H, WARP_H, A + (long)(act::WH + (i - 1) * 128) * P, WARP_H, WARP_H, G(F.warp_w[i]), ldw, dl);
            add_bias(dl, F.warp_b[i], WARP_H);
            if (i == 4) {
                b.tn(cur, WARP_H, WARP_H, A + (long)(act::E) * P, 16 * KB_XYZ, D_XYZ, G(F.warp_w[4]) + WARP_H, ldw);
                consts(F.warp_w[4], ldw, WARP_H, WARP_H + D_XYZ, D_DRV, dl, drv, d_drv);
                consts(F.warp_w[4], ldw, WARP_H, WARP_H + D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
            }
            b.nn(cur, WARP_H, WARP_H, W(F.warp_w[i]), ldw, WARP_H, nxt, WARP_H, 0, A + (long)(act::WH + (i - 1) * 128) * P, WARP_H, 0.0f);
            float *t = cur; cur = nxt; nxt = t;
        }
        float *dl = newdb(WARP_H);
        b.tn(cur, WARP_H, WARP_H, A + (long)(act::E) * P, 16 * KB_XYZ, D_XYZ, G(F.warp_w[0]), D_DEF_IN, dl);
        add_bias(dl, F.warp_b[0], WARP_H);
        consts(F.warp_w[0], D_DEF_IN, WARP_H, D_XYZ, D_DRV, dl, drv, d_drv);
        consts(F.warp_w[0], D_DEF_IN, WARP_H, D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
    }
#else
    (void)drv; (void)d_p36; (void)p36; (void)g3;   // no deformation nets: the gradient stops at the (input) point
#endif
    (void)layer_params;
    if (!b.dry) b.flush_deferred();
    if (!b.err && dbo > DB_SCRATCH - 64) b.err = (int)hipErrorOutOfMemory;
    return b.err;
    };      // walk
    b.dry = true;
    if (int e = walk()) return e;
    b.flush_copies();
    if (b.err) return b.err;
    b.dry = false;
    return walk();
}

#if SAHS_MODEL == 0
// ================================================================================================================================
// The fused walk (round 4; AudioFaceModel, split-operand arithmetic): per part of the field TWO GEMM-class launches instead of ~38 --
//   1. field_backward_chain_{rad,def}_kernel (field_bwd_chain.hip): the whole data-gradient chain, sample-major, every layer's dZ stored once
//      into its plane of `dact` (the act:: layout), masks from the sign bits the saving forward wrote;
//   2. gemm_tn_jobs_kernel: every layer's dW (+ db) = dZ^T X from those planes and the saved activations, one job table;
// around them the small kernels of the per-layer walk (feature-grid scatter, encodings, per-frame-constant columns, deferred adds).
// part 1: deformation nets (seam gradient xwg_in (P,8) -> parameters); part 2: radiance nets of `level` (d_raw (P,16) -> parameters, seam
// gradient to xwg_out); part 3: part 2, then part 1 on (its seam gradient + xwg_in).  actbuf / bits: the buffers the saving forward of that
// part wrote (a plane of column c at actbuf + c * P; for part 3 bits = [deformation planes | radiance planes]).
// ================================================================================================================================
extern "C" {
long sahs_bwd_chain_stream_hw(int part);
int sahs_bwd_chain_pack_launch(const float *flat, void *stream_out, int level, int part, hipStream_t stream);
int sahs_bwd_chain_rad_launch(const void *bstream, long P, const float *d_raw, const uint32_t *bits, float *dact, float *dgridf, float *din_a,
                              float *din_b, int num_cu, hipStream_t stream);
int sahs_bwd_chain_def_launch(const void *bstream, long P, const float *xwg, const float *actbuf, const uint32_t *bits, float *dact, float *g3,
                              float *dw4, int num_cu, hipStream_t stream);
// field_bwd_chain_f32.hip: the same chains in exact fp32 products
long sahs_bwd_chain_f32_stream_floats(int part);
int sahs_bwd_chain_f32_pack_launch(const float *flat, float *stream_out, int level, int part, hipStream_t stream);
int sahs_bwd_chain_f32_rad_launch(const float *bstream, long P, const float *d_raw, const uint32_t *bits, float *dact, float *dgridf, float *din_a,
                                  float *din_b, int num_cu, hipStream_t stream);
int sahs_bwd_chain_f32_def_launch(const float *bstream, long P, const float *xwg, const float *actbuf, const uint32_t *bits, float *dact, float *g3,
                                  float *dw4, int num_cu, hipStream_t stream);
}

namespace {
constexpr long RAD_PLANES = act::STRIDE - act::XW, DEF_PLANES = act::XW;        // floats per sample of the dZ planes of a part
struct FusedWs {      // workspace of one part, in floats
    // the part's transposed weight stream: split bf16 (hi + lo halfwords) or fp32, whichever the walk's arithmetic is
    static long stream(int part) { const long a = sahs_bwd_chain_stream_hw(part) / 2, b = sahs_bwd_chain_f32_stream_floats(part); return a > b ? a : b; }
    static long rad(long P) { return P * (RAD_PLANES + 32 + 2 * DIN_LD + 8) + DB_SCRATCH + 2 * GRID_FLOATS + stream(2) + HEAD_FLOATS; }
    static long def(long P) { return P * (DEF_PLANES + 8) + DB_SCRATCH + stream(1); }
};

__global__ void add_rows8_kernel(long n, const float *__restrict__ a, float *__restrict__ y)      // y[i] += a[i] over (P,8) rows, as float4s
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<f32x4 *>(y)[i];
        const f32x4 u = reinterpret_cast<const f32x4 *>(a)[i];
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        reinterpret_cast<f32x4 *>(y)[i] = v;
    }
}

// the job tables of one part -> one launch each: the wide layers (M, N in (128, 256]: whole 256 x 256 blocks per workgroup) and the rest
struct TnList {
    TnBatch b, w; int n = 0, tiles = 0, nw = 0;
    void add(const float *dY, long ldy, int M, const float *X, long ldx, int N, float *dW, long ldw, float *db = nullptr)
    {
        static const bool no_wide = getenv("SAHS_BWD_TN_NOWIDE") != nullptr;      // (A/B aid: everything through the 128 x 128 tiles)
        if (!no_wide && M <= 256 && N <= 256 && M >= 128 && N >= 128 && (M > 128 || N > 128)) {
            if (nw < MAX_TN_JOBS) w.j[nw] = TnJob{dY, X, dW, db, ldy, ldx, ldw, M, N};
            ++nw;
            return;
        }
        if (n < MAX_TN_JOBS) b.j[n] = TnJob{dY, X, dW, db, ldy, ldx, ldw, M, N};
        ++n;
        tiles += ((N + GT - 1) / GT) * ((M + GT - 1) / GT);
    }
    int launch(long P, const float *zero, int num_cu, hipStream_t st, bool f32)
    {
        if (n > MAX_TN_JOBS || nw > MAX_TN_JOBS) return (int)hipErrorOutOfMemory;
        if (f32) return launch_f32(P, zero, num_cu, st);
        static sahs_once::Flags attr_set, attr_set_w;
        hipError_t ae = sahs_once::per_device(attr_set, [&]() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_jobs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
        });
        if (ae != hipSuccess) return (int)ae;
        ae = sahs_once::per_device(attr_set_w, [&]() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_jobs256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES);
        });
        if (ae != hipSuccess) return (int)ae;
        // ranges of >= 1024 samples (a multiple of 16), `rounds` rounds of equal items per workgroup: fewer, longer items mean fewer atomic
        // epilogues (measured on the training step: wide kernel 1.21 / 1.24 / 1.27 / 1.31 ms at 1 / 2 / 3 / 5 rounds; the narrow one is flat)
        static const long rounds_env = getenv("SAHS_BWD_TN_ROUNDS") ? atol(getenv("SAHS_BWD_TN_ROUNDS")) : 0;      // (tuning aid)
        auto range_for = [&](long workgroups, int units, long rounds) {
            long nsplit = (rounds_env > 0 ? rounds_env : rounds) * workgroups / (units > 0 ? units : 1);
            if (nsplit < 1) nsplit = 1;
            long range = ((P + nsplit - 1) / nsplit + 15) / 16 * 16;
            return range < 1024 ? 1024L : range;
        };
        if (nw > 0) {
            gemm_tn_jobs256_kernel<<<num_cu, TN_THREADS, TW_LDS_BYTES, st>>>(w, nw, P, range_for(num_cu, nw, 1), zero);      // one 128-KB workgroup per CU
            if (hipGetLastError() != hipSuccess) return (int)hipErrorLaunchFailure;
        }
        if (n > 0) {
            const int G = 2 * num_cu / 8 * 8;                     // two 80-KB workgroups per CU
            gemm_tn_jobs_kernel<<<G, TN_THREADS, TN_LDS_BYTES, st>>>(b, n, tiles, P, range_for(G, tiles, 2), zero);
        }
        return (int)hipGetLastError();
    }
    // units with relative K-step costs -> ranges per unit so that `slots` items of equal cost come out (ranges of >= 1024 samples)
    static bool make_plan(TnPlan &pl, const float *cost, int nunits, long slots, long P)
    {
        if (nunits > MAX_TN_UNITS) return false;
        double total = 0.0;
        for (int i = 0; i < nunits; ++i) total += cost[i];
        const long nmax = P / 1024 > 0 ? P / 1024 : 1;
        long n[MAX_TN_UNITS], sum = 0;
        for (int i = 0; i < nunits; ++i) {
            n[i] = (long)(cost[i] / total * (double)slots + 0.5);
            n[i] = n[i] < 1 ? 1 : (n[i] > nmax ? nmax : n[i]);
            sum += n[i];
        }
        while (sum > slots) {      // never one item more than the slots: it would be a whole extra round of the launch (measured: 1.9 -> 3.5 ms)
            int big = 0;
            for (int i = 1; i < nunits; ++i)
                if (n[i] > n[big]) big = i;
            if (n[big] <= 1) break;
            --n[big]; --sum;
        }
        int start = 0;
        for (int i = 0; i < nunits; ++i) {
            if (start + n[i] > 65535) return false;
            pl.u[i].start = (unsigned short)start;
            pl.u[i].n = (unsigned short)n[i];
            start += (int)n[i];
        }
        pl.nunits = nunits;
        pl.items = start;
        return true;
    }
    int launch_f32(long P, const float *zero, int num_cu, hipStream_t st)
    {
        static sahs_once::Flags attr_set, attr_set_w;
        hipError_t ae = sahs_once::per_device(attr_set, [&]() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_jobs_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_BYTES);
        });
        if (ae != hipSuccess) return (int)ae;
        ae = sahs_once::per_device(attr_set_w, [&]() {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_jobs256_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TFW_LDS_BYTES);
        });
        if (ae != hipSuccess) return (int)ae;
        static const float c0w = getenv("SAHS_TNF_C0W") ? (float)atof(getenv("SAHS_TNF_C0W")) : 0.27f;      // (tuning aids: a K-step's fixed part in the two
        static const float c0n = getenv("SAHS_TNF_C0N") ? (float)atof(getenv("SAHS_TNF_C0N")) : 0.4f;      //  cost models, items per workgroup of the narrow launch)
        static const long rounds_n = getenv("SAHS_TNF_ROUNDS") ? atol(getenv("SAHS_TNF_ROUNDS")) : 3;
        if (nw > 0) {      // one 128-KB workgroup per CU, one item each
            TnPlan pl;
            float cost[MAX_TN_UNITS];
            for (int i = 0; i < nw && i < MAX_TN_UNITS; ++i) {
                cost[i] = c0w + (w.j[i].M > 128 ? 2.0f : 1.0f);      // (rows past M are the zero page: the waves that own them skip their MFMAs; 0.27: a K-step's fixed part)
                pl.u[i].job = (unsigned char)i; pl.u[i].bx = pl.u[i].by = pl.u[i].pad = 0;
            }
            if (!make_plan(pl, cost, nw, num_cu, P)) return (int)hipErrorOutOfMemory;
            gemm_tn_jobs256_f32_kernel<<<num_cu, TN_THREADS, TFW_LDS_BYTES, st>>>(w, pl, P, zero);
            if (hipGetLastError() != hipSuccess) return (int)hipErrorLaunchFailure;
        }
        if (n > 0) {       // two 64-KB workgroups per CU, three items each (measured on the training step: 0.91 / 0.69 / 0.64 / 0.64 ms per launch at 1 / 2 / 3 / 4)
            const int G = 2 * num_cu;
            TnPlan pl;
            float cost[MAX_TN_UNITS];
            int nu = 0;
            for (int i = 0; i < n; ++i) {
                const int nx = (b.j[i].N + GT - 1) / GT, ny = (b.j[i].M + GT - 1) / GT;
                for (int by = 0; by < ny; ++by)
                    for (int bx = 0; bx < nx; ++bx, ++nu) {
                        if (nu >= MAX_TN_UNITS) return (int)hipErrorOutOfMemory;
                        const int mt = b.j[i].M - by * GT < GT ? b.j[i].M - by * GT : GT, nt = b.j[i].N - bx * GT < GT ? b.j[i].N - bx * GT : GT;
                        // a K-step: its fixed part (barrier, DMA issue, first LDS round trip; fitted on the training step, tools/sweep_tnf.sh: 0.54 / 0.54 /
                        // 0.55 / 0.56 / 0.57 ms per launch at 0.4 / 0.7 / 1.0 / 1.3 / 1.8) + the tile's valid accumulators
                        cost[nu] = c0n + (float)(((mt + 15) / 16) * ((nt + 15) / 16)) / 64.0f;
                        pl.u[nu].job = (unsigned char)i; pl.u[nu].bx = (unsigned char)bx; pl.u[nu].by = (unsigned char)by; pl.u[nu].pad = 0;
                    }
            }
            if (!make_plan(pl, cost, nu, rounds_n * G, P)) return (int)hipErrorOutOfMemory;
            gemm_tn_jobs_f32_kernel<<<G, TN_THREADS, TF_LDS_BYTES, st>>>(b, pl, P, zero);
        }
        return (int)hipGetLastError();
    }
};
}  // namespace

extern "C" long sahs_field_backward_fused_ws_words(int part, long P)
{
    return part == 1 ? FusedWs::def(P) : (part == 2 ? FusedWs::rad(P) : FusedWs::rad(P) + FusedWs::def(P) + P * 8);
}

static int fused_rad(const float *flat, const float *frame, int level, long P, const float *actbuf, const uint32_t *bits, const float *d_raw,
                     float *xwg_out, float *grad_flat, float *grad_cond, float *ws, int num_cu, hipStream_t stream)
{
    Bwd b{stream, P};
    const FlatOffsets &F = kFlat;
    const FlatOffsets::Lvl &Lv = F.lvl[level];
    float *dact_mem = ws, *dgridf = dact_mem + P * RAD_PLANES, *din_a = dgridf + P * 32, *din_b = din_a + P * DIN_LD, *dxw = din_b + P * DIN_LD,
          *dw = dxw + P * 4, *db = dw + P * 4, *grid_cl = db + DB_SCRATCH, *dgrid_cl = grid_cl + GRID_FLOATS, *bstream = dgrid_cl + GRID_FLOATS,
          *heads = bstream + FusedWs::stream(2);
    float *dact = dact_mem - (long)act::XW * P;              // plane of act:: column c at dact + c * P (columns >= XW are backed)
    b.zero = db + DB_SCRATCH - 64;
    if (hipMemsetAsync(db, 0, sizeof(float) * DB_SCRATCH, stream) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(dgrid_cl, 0, sizeof(float) * GRID_FLOATS, stream) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(heads, 0, sizeof(float) * HEAD_FLOATS, stream) != hipSuccess) return (int)hipGetLastError();
    const bool f32 = !Bwd::x3();      // exact fp32 products (ops.backward_gemm_precision("fp32")): the fp32 chain and job kernels
    int e = f32 ? sahs_bwd_chain_f32_pack_launch(flat, bstream, level, 2, stream) : sahs_bwd_chain_pack_launch(flat, bstream, level, 2, stream);
    if (e) return e;
    e = f32 ? sahs_bwd_chain_f32_rad_launch(bstream, P, d_raw, bits, dact, dgridf, din_a, din_b, num_cu, stream)
            : sahs_bwd_chain_rad_launch(bstream, P, d_raw, bits, dact, dgridf, din_a, din_b, num_cu, stream);
    if (e) return e;
    // ---- encodings + feature grid -> the seam gradient (what the deformation part waits for) ----
    {
        const int tb = (int)(GRID_FLOATS / 32 / 32);
        grid_transpose_kernel<<<tb, 256, 0, stream>>>(flat + F.grid, grid_cl, 0); b.check();
        const long gb = (P + 127) / 128;
        grid_backward_kernel<<<(unsigned)(gb < 8192 ? gb : 8192), 256, 0, stream>>>(P, actbuf, dgridf, grid_cl, dgrid_cl, dxw); b.check();
        grid_transpose_kernel<<<tb, 256, 0, stream>>>(dgrid_cl, grad_flat + F.grid, 1); b.check();
        encode_backward_kernel<<<2048, 256, 0, stream>>>(P, actbuf, din_a, din_b, dxw, dw, xwg_out); b.check();
    }
    // ---- every weight gradient of the part: one job table ----
    const float *A = actbuf;
    auto AC = [&](int c) { return A + (long)c * P; };
    auto DA = [&](int c) { return dact + (long)c * P; };
    auto G = [&](long off) { return grad_flat + off; };
    const float *p36 = frame + FRAME_POSE_OFF, *drv = frame + FRAME_DRV_OFF;
    const float *trc = TRUNK_SEES_POSE ? p36 : drv;
    float *d_trc = grad_cond + (TRUNK_SEES_POSE ? 80 : 0);
    int dbo = 0;
    auto scratch_db = [&](int n) { float *p = db + dbo; dbo += (n + 3) / 4 * 4; return p; };
    auto defer_add = [&](const float *src, float *dst, int n) { if (b.naxpy < MAX_AXPY_JOBS) b.axpys.j[b.naxpy++] = AxpyJob{src, dst, n}; else if (!b.err) b.err = (int)hipErrorOutOfMemory; };
    auto defer_consts = [&](long woff, long ld, int rows, int col0, int cols, const float *dbl, const float *c, float *dc) {
        if (b.nconst < MAX_CONST_JOBS) {
            b.consts.j[b.nconst++] = ConstJob{flat + woff, G(woff), dbl, c, dc, ld, rows, cols, col0};
            b.const_maxcols = cols > b.const_maxcols ? cols : b.const_maxcols;
        } else if (!b.err) b.err = (int)hipErrorOutOfMemory;
    };
    TnList L;
    // the three heads read the whole d_raw row against 16-row scratch gradients (rows 3..14 fc_seg, 0..2 fc_rgb, 15 fc_alpha)
    L.add(d_raw, 16, 16, AC(act::S + 384), BR_H, BR_H, heads + HEAD_G_SEG, BR_H, heads + HEAD_DB);
    L.add(d_raw, 16, 16, AC(act::C + 384), BR_H, BR_H, heads + HEAD_G_RGB, BR_H);
    L.add(d_raw, 16, 16, AC(act::FEAT), TR_H, TR_H, heads + HEAD_G_ALPHA, TR_H);
    defer_add(heads + HEAD_G_SEG + 3 * BR_H, G(Lv.segout_w), N_SEG * BR_H);
    defer_add(heads + HEAD_DB + 3, G(Lv.segout_b), N_SEG);
    defer_add(heads + HEAD_DB, G(Lv.rgb_b), 3);
    defer_add(heads + HEAD_DB + 15, G(Lv.alpha_b), 1);
    defer_add(heads + HEAD_G_RGB, G(Lv.rgb_w), 3 * BR_H);
    defer_add(heads + HEAD_G_ALPHA + 15 * TR_H, G(Lv.alpha_w), TR_H);
    for (int i = 3; i >= 1; --i) {
        L.add(DA(act::S + 128 * i), BR_H, BR_H, AC(act::S + 128 * (i - 1)), BR_H, BR_H, G(Lv.seg_w[i]), BR_H, G(Lv.seg_b[i]));
        L.add(DA(act::C + 128 * i), BR_H, BR_H, AC(act::C + 128 * (i - 1)), BR_H, BR_H, G(Lv.dir_w[i]), BR_H, G(Lv.dir_b[i]));
    }
    L.add(DA(act::S), BR_H, BR_H, AC(act::FEAT), TR_H, TR_H, G(Lv.seg_w[0]), TR_H, G(Lv.seg_b[0]));
    L.add(DA(act::C), BR_H, BR_H, AC(act::FEAT), TR_H, TR_H, G(Lv.dir_w[0]), D_DIR_IN, G(Lv.dir_b[0]));
    L.add(DA(act::C), BR_H, BR_H, AC(act::DIR), 32, D_DIR, G(Lv.dir_w[0]) + TR_H, D_DIR_IN);
    L.add(DA(act::C), BR_H, BR_H, AC(act::GRID), 32, D_GRID, G(Lv.dir_w[0]) + TR_H + D_DIR, D_DIR_IN);
    L.add(DA(act::FEAT), TR_H, TR_H, AC(act::T + (TR_LAYERS - 1) * 256), TR_H, TR_H, G(Lv.feat_w), TR_H, G(Lv.feat_b));
    for (int i = TR_LAYERS - 1; i >= 1; --i) {
        const long ldw = (i == 3) ? TR_H + D_TR_IN : TR_H;
        float *dl = (i == 3) ? scratch_db(TR_H) : G(Lv.xyz_b[i]);
        L.add(DA(act::T + 256 * i), TR_H, TR_H, AC(act::T + 256 * (i - 1)), TR_H, TR_H, G(Lv.xyz_w[i]), ldw, dl);
        if (i == 3) {
            L.add(DA(act::T + 768), TR_H, TR_H, AC(act::PEX), 16 * KB_XYZ, D_XYZ, G(Lv.xyz_w[3]) + TR_H, ldw);
            if (D_AMB > 0) L.add(DA(act::T + 768), TR_H, TR_H, AC(act::PEW), 16 * KB_AMB, D_AMB, G(Lv.xyz_w[3]) + TR_H + D_XYZ, ldw);
            defer_add(dl, G(Lv.xyz_b[3]), TR_H);
            defer_consts(Lv.xyz_w[3], ldw, TR_H, TR_H + D_XYZ + D_AMB, D_TR_CONST, dl, trc, d_trc);
        }
    }
    {
        float *dl = scratch_db(TR_H);
        L.add(DA(act::T), TR_H, TR_H, AC(act::PEX), 16 * KB_XYZ, D_XYZ, G(Lv.xyz_w[0]), D_TR_IN, dl);
        if (D_AMB > 0) L.add(DA(act::T), TR_H, TR_H, AC(act::PEW), 16 * KB_AMB, D_AMB, G(Lv.xyz_w[0]) + D_XYZ, D_TR_IN);
        defer_add(dl, G(Lv.xyz_b[0]), TR_H);
        defer_consts(Lv.xyz_w[0], D_TR_IN, TR_H, D_XYZ + D_AMB, D_TR_CONST, dl, trc, d_trc);
    }
    if (b.err) return b.err;
    e = L.launch(P, b.zero, num_cu, stream, f32);
    if (e) return e;
    b.flush_deferred();
    return b.err;
}

static int fused_def(const float *flat, const float *frame, long P, const float *actbuf, const uint32_t *bits, const float *xwg, float *grad_flat,
                     float *grad_cond, float *ws, int num_cu, hipStream_t stream)
{
    Bwd b{stream, P};
    const FlatOffsets &F = kFlat;
    float *dact = ws, *g3 = dact + P * DEF_PLANES, *dw4 = g3 + P * 4, *db = dw4 + P * 4, *bstream = db + DB_SCRATCH;
    b.zero = db + DB_SCRATCH - 64;
    if (hipMemsetAsync(db, 0, sizeof(float) * DB_SCRATCH, stream) != hipSuccess) return (int)hipGetLastError();
    const bool f32 = !Bwd::x3();
    int e = f32 ? sahs_bwd_chain_f32_pack_launch(flat, bstream, 0, 1, stream) : sahs_bwd_chain_pack_launch(flat, bstream, 0, 1, stream);
    if (e) return e;
    e = f32 ? sahs_bwd_chain_f32_def_launch(bstream, P, xwg, actbuf, bits, dact, g3, dw4, num_cu, stream)
            : sahs_bwd_chain_def_launch(bstream, P, xwg, actbuf, bits, dact, g3, dw4, num_cu, stream);
    if (e) return e;
    auto AC = [&](int c) { return actbuf + (long)c * P; };
    auto DA = [&](int c) { return dact + (long)c * P; };
    auto G = [&](long off) { return grad_flat + off; };
    const float *p36 = frame + FRAME_POSE_OFF, *drv = frame + FRAME_DRV_OFF;
    float *d_drv = grad_cond + 0, *d_p36 = grad_cond + 80;
    int dbo = 0;
    auto scratch_db = [&](int n) { float *p = db + dbo; dbo += (n + 3) / 4 * 4; return p; };
    auto defer_add = [&](const float *src, float *dst, int n) { if (b.naxpy < MAX_AXPY_JOBS) b.axpys.j[b.naxpy++] = AxpyJob{src, dst, n}; else if (!b.err) b.err = (int)hipErrorOutOfMemory; };
    auto defer_consts = [&](long woff, long ld, int rows, int col0, int cols, const float *dbl, const float *c, float *dc) {
        if (b.nconst < MAX_CONST_JOBS) {
            b.consts.j[b.nconst++] = ConstJob{flat + woff, G(woff), dbl, c, dc, ld, rows, cols, col0};
            b.const_maxcols = cols > b.const_maxcols ? cols : b.const_maxcols;
        } else if (!b.err) b.err = (int)hipErrorOutOfMemory;
    };
    TnList L;
    auto net = [&](const float *head_dy, int head_rows, long fw, long fb, const long *w, const long *bs, int Hn, int col) {
        L.add(head_dy, 4, head_rows, AC(col + 5 * Hn), Hn, Hn, G(fw), Hn, G(fb));
        for (int i = 5; i >= 1; --i) {
            const long ldw = (i == 4) ? Hn + D_DEF_IN : Hn;
            float *dl = (i == 4) ? scratch_db(Hn) : G(bs[i]);
            L.add(DA(col + i * Hn), Hn, Hn, AC(col + (i - 1) * Hn), Hn, Hn, G(w[i]), ldw, dl);
            if (i == 4) {
                L.add(DA(col + 4 * Hn), Hn, Hn, AC(act::E), 16 * KB_XYZ, D_XYZ, G(w[4]) + Hn, ldw);
                defer_add(dl, G(bs[4]), Hn);
                defer_consts(w[4], ldw, Hn, Hn + D_XYZ, D_DRV, dl, drv, d_drv);
                defer_consts(w[4], ldw, Hn, Hn + D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
            }
        }
        float *dl = scratch_db(Hn);
        L.add(DA(col), Hn, Hn, AC(act::E), 16 * KB_XYZ, D_XYZ, G(w[0]), D_DEF_IN, dl);
        defer_add(dl, G(bs[0]), Hn);
        defer_consts(w[0], D_DEF_IN, Hn, D_XYZ, D_DRV, dl, drv, d_drv);
        defer_consts(w[0], D_DEF_IN, Hn, D_XYZ + D_DRV, D_POSE, dl, p36, d_p36);
    };
    net(dw4, AMB_DIM, F.hyp_fw, F.hyp_fb, F.hyp_w, F.hyp_b, HYP_H, act::HH);
    net(g3, 3, F.warp_fw, F.warp_fb, F.warp_w, F.warp_b, WARP_H, act::WH);
    if (b.err) return b.err;
    e = L.launch(P, b.zero, num_cu, stream, f32);
    if (e) return e;
    b.flush_deferred();
    return b.err;
}

extern "C" int sahs_field_backward_fused_launch(const float *flat, const float *frame, int level, int part, long P, const float *actbuf,
                                                const uint32_t *bits, const float *d_raw, const float *xwg_in, float *xwg_out, float *grad_flat,
                                                float *grad_cond, float *ws, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    if (P > 4000000L) return -3;      // 32-bit byte offsets inside a plane (256 floats per sample)
    if (part == 1) return fused_def(flat, frame, P, actbuf, bits, xwg_in, grad_flat, grad_cond, ws, num_cu, stream);
    if (part == 2) return fused_rad(flat, frame, level, P, actbuf, bits, d_raw, xwg_out, grad_flat, grad_cond, ws, num_cu, stream);
    if (part != 3) return -2;
    // everything: the radiance part's seam gradient (+ xwg_in, the fine pass's share when the deformation was shared) feeds the deformation part
    float *seam = ws + FusedWs::rad(P), *ws_def = seam + P * 8;
    int e = fused_rad(flat, frame, level, P, actbuf, bits + (long)sbits::BD_WORDS * P, d_raw, seam, grad_flat, grad_cond, ws, num_cu, stream);
    if (e) return e;
    if (xwg_in != nullptr) {
        add_rows8_kernel<<<2048, 256, 0, stream>>>(P * 2, xwg_in, seam);
        if ((e = (int)hipGetLastError())) return e;
    }
    return fused_def(flat, frame, P, actbuf, bits, seam, grad_flat, grad_cond, ws_def, num_cu, stream);
}
#endif      // SAHS_MODEL == 0
