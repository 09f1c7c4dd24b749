// Exhaustive check of the packed-integer LeakyReLU of field_bf16w.hip (pack_tick, default build) on the GPU: every bf16 bit pattern in
// both halves of a dword through  s = v_pk_ashrrev_i16(15, d);  r = v_pk_mad_i16(s, K, d) clamp  against the intended function
//   f(p) = p                                   for p >= 0 (as int16: positive values, +0, +inf, +NaN pass through)
//        = 0x8000 | max((p & 0x7fff) - K, 0)   for p <  0 (the magnitude pattern reduced by K = 850, saturating at -0.0).
// hipcc --offload-arch=gfx950 -O2 tools/micro/pattern_leaky_check.hip -o /tmp/plc && /tmp/plc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k(uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. 65535
    const uint32_t d = (i & 0xffffu) | ((~i & 0xffffu) << 16);          // pattern i in the low half, its complement in the high half
    const uint32_t kk = 0x03520352u;
    uint32_t s, r;
    asm volatile("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(s) : "v"(d));
    asm volatile("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(s), "s"(kk), "v"(d));
    out[i] = r;
}

static uint32_t f(uint32_t p)
{
    if (!(p & 0x8000u)) return p;
    const int m = (int)(p & 0x7fffu) - 850;
    return 0x8000u | (uint32_t)(m > 0 ? m : 0);
}

int main()
{
    uint32_t *d;
    if (hipMalloc(&d, 65536 * 4) != hipSuccess) return 2;
    k<<<256, 256>>>(d);
    std::vector<uint32_t> h(65536);
    if (hipMemcpy(h.data(), d, 65536 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (uint32_t i = 0; i < 65536; ++i) {
        const uint32_t lo = i, hi = ~i & 0xffffu, want = f(lo) | (f(hi) << 16);
        if (h[i] != want && bad++ < 8) printf("pattern %04x|%04x: got %08x want %08x\n", hi, lo, h[i], want);
    }
    printf("pattern_leaky_check: %d of 65536 dwords differ\n", bad);
    return bad ? 1 : 0;
}
