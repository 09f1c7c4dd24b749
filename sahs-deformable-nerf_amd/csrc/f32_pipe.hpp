// f32_pipe.hpp -- the fp32 layer pipeline shared by the forward field kernel (field_f32.hip) and the fp32 backward chains
// (field_bwd_chain_f32.hip): the weight-stream context (LDS-DMA double buffer) and one dense layer on v_mfma_f32_16x16x4_f32 whose D tile
// (lane (q, j): features 4q..4q+3 of point j) is the next layer's B operand.  What happens to a finished tile -- bias + activation +
// optional save (forward), derivative mask from the sign bits + store of dZ (backward) -- is an epilogue policy.
#pragma once
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"

namespace SAHS_NS {

constexpr int F32_THREADS = 512;
constexpr int F32_PTS_PER_WAVE = 16;
constexpr int F32_PTS_PER_WG = (F32_THREADS / WAVE) * F32_PTS_PER_WAVE;   // 128
constexpr int DBG_STRIDE = 56;   // test seam: [dx3, w2, T0[0], trunk layers 1..8 [0], D0[0], D3[0], S0[0], S3[0], pad]
constexpr int LDS_BUF_FLOATS = CHUNK_FLOATS_MAX;                           // 32 KB each, two of them
constexpr int LDS_BIAS_OFF = 2 * LDS_BUF_FLOATS;
constexpr int LDS_FLOATS = LDS_BIAS_OFF + ((BIAS_FLOATS + 3) / 4) * 4;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;
typedef const __attribute__((address_space(3))) float *lds_cfloat;
typedef const __attribute__((address_space(3))) f32x4 *lds_cf4;

constexpr int LDS_STASH_OFF = LDS_BIAS_OFF + ((BIAS_FLOATS + 3) / 4) * 4;     // floats
constexpr int STASH_FLOATS = 8;                                                // per sample: x'[3], w[2] (+pad)
constexpr int LDS_TOTAL_FLOATS = LDS_STASH_OFF + F32_PTS_PER_WG * STASH_FLOATS;
static_assert(LDS_TOTAL_FLOATS * 4 <= 160 * 1024, "LDS budget");
constexpr int PIECE_FLOATS = F32_THREADS * 4;                                  // one DMA piece = 8 KB (1 KB per wave)
constexpr int MAX_PIECES = LDS_BUF_FLOATS / PIECE_FLOATS;                      // 4
constexpr int A_AHEAD = 3;                                                     // A fragments (one ds_read_b128 = 4 MFMAs) in flight

struct Ctx {
    const float *stream;      // this level's packed weight stream (global)
    float *lds;               // dynamic LDS base
    int buf;                  // LDS buffer (0/1) holding the current chunk; toggles per chunk (the chunk count is odd, the stream wraps)
    int lane, q, wave;
    // Next chunk's LDS-DMA.  The chunk sequence is static, so its offset is tracked arithmetically (no table load on the
    // critical path) and its size -- hence its number of 8-KB pieces -- is a compile-time constant at every call site.
    uint32_t off;             // float offset of the next chunk to prefetch (uniform)
    uint32_t wrap_at, wrap_to; // the part of the stream this launch walks: [wrap_to, wrap_at) (whole stream, deformation nets only, or radiance net only)
    const f32x4 *nx_src; f32x4 *nx_dst;
    lds_cfloat bias_lane;     // LDS address of this lane's bias rows (refreshed opaquely per sample tile: see refresh())

    __device__ __forceinline__ void begin_chunk(int next_floats)
    {
        if (off >= wrap_at) off = wrap_to;
        nx_src = reinterpret_cast<const f32x4 *>(stream + off) + lane;
        nx_dst = reinterpret_cast<f32x4 *>(lds + (buf ^ 1) * LDS_BUF_FLOATS);
        off += (uint32_t)next_floats;
    }
    __device__ __forceinline__ void issue_piece(int p)   // global_load_lds writes LDS at (wave-uniform base + lane*16)
    {
        const int base = p * F32_THREADS + wave * WAVE;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(nx_src + base), (lds_ptr_t)(nx_dst + base), 16, 0, 0);
    }
    // N: the wave's vector-memory operations that may still be in flight behind the chunk's closing barrier -- the stores of its last
    // finished tiles, all younger than the chunk's LDS-DMA pieces (vmcnt retires in issue order: "all but the N youngest done" covers every
    // piece).  N = 0: everything drains (the forward without saved activations: __syncthreads, as before).
    template <int N = 0>
    __device__ __forceinline__ void end_chunk()
    {
        if constexpr (N == 0) {
            __syncthreads();   // drains the in-flight global_load_lds (vmcnt(0)) and orders buffer reuse
        } else {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
        }
        buf ^= 1;
    }
    __device__ __forceinline__ const f32x4 *cur() const { return reinterpret_cast<const f32x4 *>(lds + buf * LDS_BUF_FLOATS); }
    // Without the opaque refresh LICM precomputes every tile's bias address outside the persistent loop (~280 VGPRs' worth),
    // spills them, and each reload (a scratch = VMEM op) then waits on vmcnt -- i.e. on the LDS-DMA weight prefetch.
    __device__ __forceinline__ void refresh()
    {
        uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)lds) + (LDS_BIAS_OFF + 4 * q) * 4;
        asm volatile("" : "+v"(a));
        bias_lane = (lds_cfloat)(uintptr_t)a;
    }
    __device__ __forceinline__ f32x4 bias4(int off_) const { return *reinterpret_cast<lds_cf4>(bias_lane + off_); }
};

__device__ __forceinline__ f32x4 act4(f32x4 v, float slope)   // slope in [0,1]: leaky relu == max(x, slope*x)
{
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = fmaxf(v[r], v[r] * slope);
    return o;
}

// Vector-memory operations a wave is GUARANTEED to have issued after the last LDS-DMA piece of a chunk: per_tile stores for every tile that
// finishes at or behind that piece's step (a finished tile's store follows the piece of the same step in program order; a scheduling barrier
// behind the last piece keeps it there).  0 when pieces are still issued behind the step loop (short chunks).  A saving forward waited for
// vmcnt(0) at every chunk end -- i.e. for the acknowledgement of stores it had issued a few hundred cycles earlier, ~100 times per sample tile.
constexpr int stores_behind_last_piece(int STEPS, int PAIR, int KB, int next_floats, int per_tile)
{
    const int npieces = (next_floats + PIECE_FLOATS - 1) / PIECE_FLOATS;
    const int pstep = (STEPS / 2 >= npieces) ? (STEPS / 2) / npieces : 1;
    if (per_tile == 0 || npieces > (STEPS + pstep - 1) / pstep) return 0;
    int n = 0;
    for (int s = (npieces - 1) * pstep; s < STEPS; ++s)
        if ((s % (PAIR * KB)) / PAIR == KB - 1) n += per_tile;
    return n > 48 ? 48 : n;
}

// One dense layer.  in0[KB0] ++ in1[KB1] are the input k-blocks (16 features each); out[NT] the 16-row output tiles.
// NEXT = floats in the chunk that follows this layer's last chunk.
// A chunk is a flat run of G*KB steps; step = one A fragment (ds_read_b128) feeding 4 MFMAs.  Tiles are taken in pairs so
// two independent accumulation chains alternate (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency).
// EP: first(cx, out, t) = the accumulator a tile starts from; done<NT>(acc, out, t) = what becomes of the finished tile; EP::kStores = the
// vector-memory stores done() issues UNCONDITIONALLY per tile (the chunk's closing wait leaves them in flight).
template <int KB0, int KB1, int NT, int NEXT, class EP>
__device__ __forceinline__ void dense_ep(Ctx &cx, const f32x4 *in0, const f32x4 *in1, f32x4 *out, EP &ep)
{
    constexpr int KB = KB0 + KB1;
    constexpr int G = pick_G(KB, NT);
    constexpr int NCH = NT / G;
    constexpr int PAIR = (G >= 2) ? 2 : 1;             // tiles interleaved
    constexpr int STEPS = G * KB;                      // A fragments per chunk
    constexpr int N_MID = stores_behind_last_piece(STEPS, PAIR, KB, G * KB * 256, EP::kStores), N_LAST = stores_behind_last_piece(STEPS, PAIR, KB, NEXT, EP::kStores);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int nfl = (c + 1 < NCH) ? G * KB * 256 : NEXT;
        const int npieces = (nfl + PIECE_FLOATS - 1) / PIECE_FLOATS;
        // pieces of the next chunk go out during the first half of this chunk's steps
        const int pstep = (STEPS / 2 >= npieces) ? (STEPS / 2) / npieces : 1;
        cx.begin_chunk(nfl);
        const f32x4 *A = cx.cur() + cx.lane;
        // step s -> (pair p, block b, member m): order p-major, then b, then m
        auto frag = [&](int s) { const int pp = s / (PAIR * KB), r = s % (PAIR * KB); return ((pp * PAIR + r % PAIR) * KB + r / PAIR) * 64; };
        f32x4 a[A_AHEAD];
#pragma unroll
        for (int s = 0; s < A_AHEAD && s < STEPS; ++s) a[s] = A[frag(s)];
        f32x4 acc[PAIR];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int pp = s / (PAIR * KB), r = s % (PAIR * KB), m = r % PAIR, b = r / PAIR;
            const int t = c * G + pp * PAIR + m;
            if (b == 0) acc[m] = ep.first(cx, out, t);
            if (s % pstep == 0 && s / pstep < npieces) {
                cx.issue_piece(s / pstep);
                if (EP::kStores > 0 && s / pstep == npieces - 1) __builtin_amdgcn_sched_barrier(0);      // (the counted wait below: no store moves in front of the last piece)
            }
            const f32x4 x = (b < KB0) ? in0[b] : in1[b - KB0];
            const f32x4 w = a[s % A_AHEAD];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], x[k], acc[m], 0, 0, 0);
            if (s + A_AHEAD < STEPS) a[s % A_AHEAD] = A[frag(s + A_AHEAD)];
            if (b == KB - 1) ep.template done<NT>(acc[m], out, t);
        }
#pragma unroll
        for (int pc = (STEPS + pstep - 1) / pstep; pc < MAX_PIECES; ++pc)
            if (pc < npieces) cx.issue_piece(pc);
        if (c + 1 < NCH) cx.template end_chunk<N_MID>();
        else cx.template end_chunk<N_LAST>();
    }
}

// The forward epilogue.  slope: 1 = no activation, 0 = relu, 0.01 = leaky relu.  accum: start from out[] instead of the bias.
template <bool HAS_SAVE>
struct FwdEp {
    static constexpr int kStores = HAS_SAVE ? 1 : 0;
    int bias_off; bool accum; float slope;
    float *save;              // HAS_SAVE: this lane's slot of the layer's saved-activation block (never null: lanes past the end redo the last sample)
    uint32_t *bsave;          // this lane's word(s) of the layer's sign-bit plane (sahs_layout.hpp: sbits), or null
    uint32_t sgn;             // sign nibble of the even tile of a pair, until its odd partner completes the byte
    __device__ __forceinline__ f32x4 first(const Ctx &cx, const f32x4 *out, int t) const { return accum ? out[t] : cx.bias4(bias_off + 16 * t); }
    template <int NT> __device__ __forceinline__ void done(const f32x4 &acc, f32x4 *out, int t)
    {
        f32x4 o = act4(acc, slope);
        asm volatile("" : "+v"(o));     // pin: keep the finished tile from being sunk into the next layer
        out[t] = o;
        if constexpr (HAS_SAVE) *reinterpret_cast<f32x4 *>(save + 16 * t) = o;
        if (bsave != nullptr) {      // nibble t of this lane's sign word(s): bit r = value r of tile t is > 0; one byte store per tile pair
            uint32_t nib = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) nib |= (o[r] > 0.0f) ? (1u << r) : 0u;
            if ((t & 1) == 0 && t + 1 < NT) sgn = nib;
            else reinterpret_cast<unsigned char *>(bsave)[t >> 1] = (unsigned char)((t & 1) ? (sgn | (nib << 4)) : nib);
        }
    }
};

template <int KB0, int KB1, int NT, int NEXT>
__device__ __forceinline__ void dense(Ctx &cx, const f32x4 *in0, const f32x4 *in1, f32x4 *out, int bias_off, bool accum, float slope)
{
    FwdEp<false> ep{bias_off, accum, slope, nullptr, nullptr, 0u};
    dense_ep<KB0, KB1, NT, NEXT>(cx, in0, in1, out, ep);
}
// SAVING: the launch keeps its activations -- every finished tile also goes to `save` (+ its signs to bsave, where given)
template <bool SAVING, int KB0, int KB1, int NT, int NEXT>
__device__ __forceinline__ void dense_sv(Ctx &cx, const f32x4 *in0, const f32x4 *in1, f32x4 *out, int bias_off, bool accum, float slope,
                                         float *save, uint32_t *bsave = nullptr)
{
    FwdEp<SAVING> ep{bias_off, accum, slope, save, SAVING ? bsave : nullptr, 0u};
    dense_ep<KB0, KB1, NT, NEXT>(cx, in0, in1, out, ep);
}

}  // namespace SAHS_NS
