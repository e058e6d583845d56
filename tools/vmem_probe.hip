// Development probe: what one vector-memory instruction costs a compute unit when the whole chip
// is busy with them, for the access shapes of the pair kernel and the list kernels.  Every wave
// issues ROUNDS x 8 loads, eight in flight, at four waves per SIMD (the pair kernel's occupancy);
// the figure printed is  time x CUs x clock / instructions, i.e. the CU-cycles that pass per
// wave-instruction when all CUs issue them back to back.
// Measured on MI355X: gather64 50, l1set64 44, gather8 73 CU-cycles per instruction (about one lane
// per cycle whether the lines come from L1 or L2); half64 28, quarter64 22 (so about 16 cycles per
// instruction and 2.1 per active quad); gather8x4 46; far64 177; stream8 49 and stream16 99 (those two run at the
// HBM rate, 6.4 TB/s).
//   gather64 : each quad reads one 64-byte record, 16 bytes per lane (the pair kernel's fetch),
//              records near the wave's own place (hits in L2) or anywhere (misses)
//   l1set64  : each quad reads one of 256 records that stay in L1 (16 distinct records per instruction)
//   half64, quarter64 : gather64 with 8 and with 4 of the 16 quads of every wave active
//   gather8  : each lane reads its own 8-byte entry near the wave's place (the draw's gather)
//   gather8x4: gather8 with the four lanes of a quad on four consecutive entries
//   stream8  : 64 lanes read 512 contiguous bytes (the pair kernel's state-n loads)
//   stream16 : 64 lanes read 1024 contiguous bytes
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_probe tools/vmem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROUNDS = 8;  // of eight loads each
constexpr size_t SPAN = (size_t)1 << 33; // the streaming modes wrap inside 8 GiB

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0 gather64 near, 1 gather64 far, 3 gather8 near, 4 stream8, 5 stream16, 6 l1set64,
// 7 gather64 near with every other quad switched off, 8 with all quads but four switched off,
// 9 gather8 with the four lanes of a quad on four consecutive entries (32 contiguous bytes)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ buf, uint32_t nrec, uint32_t* __restrict__ sink) {
    __shared__ uint32_t hold[10240]; // 40 KB: four workgroups per CU, four waves per SIMD
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t quad = wave * 16u + (lane >> 2);
    uint32_t acc = 0;
    const uint32_t zero = sink[1]; // 0 at run time: ties a round's addresses to the round before (nothing to hoist)
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t dep = acc & zero;
        uint32_t got[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint32_t t = (uint32_t)(r * 8 + m);
            if (MODE == 7 || MODE == 8) {
                const uint32_t h = mix(quad * 64u + t);
                const uint32_t rec = (quad + (h & 2047u)) % nrec;
                got[m] = 0;
                if (MODE == 7 ? ((lane >> 2) & 1u) != 0u : (lane >> 2) < 4u) got[m] = (*(const u32x4*)(buf + (size_t)rec * 64u + 16u * q)).x;
            } else if (MODE == 0 || MODE == 1) {
                const uint32_t h = mix(quad * 64u + t);
                const uint32_t rec = MODE == 0 ? (quad + (h & 2047u)) % nrec : h % nrec; // in bounds by the modulus
                got[m] = (*(const u32x4*)(buf + (size_t)rec * 64u + 16u * q)).x;
            } else if (MODE == 6) {
                const uint32_t rec = (mix(quad * 64u + t) + dep) & 255u; // a 16 KB set: stays in L1
                got[m] = (*(const u32x4*)(buf + rec * 64u + 16u * q)).x;
            } else if (MODE == 9) {
                const uint32_t h = mix(quad * 64u + t);
                const uint32_t ent = (quad * 8u + (h & 4095u)) % (nrec * 8u - 4u) + q; // + q stays inside nrec*8 entries
                got[m] = (*(const u32x2*)(buf + (size_t)ent * 8u)).x;
            } else if (MODE == 3) {
                const uint32_t h = mix((quad * 4u + q) * 64u + t);
                const uint32_t ent = (quad * 8u + (h & 4095u)) % (nrec * 8u); // 8-byte entries: nrec*8 of them fit
                got[m] = (*(const u32x2*)(buf + (size_t)ent * 8u)).x;
            } else if (MODE == 4) {
                const size_t off = (((size_t)wave * (ROUNDS * 8) + t) * 512u) & (SPAN - 1); // + 512 stays inside SPAN
                got[m] = (*(const u32x2*)(buf + off + 8u * lane)).x;
            } else {
                const size_t off = (((size_t)wave * (ROUNDS * 8) + t) * 1024u) & (SPAN - 1); // + 1024 stays inside SPAN
                got[m] = (*(const u32x4*)(buf + off + 16u * lane)).x;
            }
        }
        acc ^= got[0] ^ got[1] ^ got[2] ^ got[3] ^ got[4] ^ got[5] ^ got[6] ^ got[7];
    }
    hold[threadIdx.x] = acc;
    if (acc == 0x12345u) sink[0] = hold[(threadIdx.x + 1) & 255]; // keeps the loads alive
}

template <int MODE>
void run(const char* name, const char* buf, uint32_t nrec, uint32_t* sink, int cus, double mhz) {
    const uint32_t blocks = 200000; // 800 000 waves, about the pair launch's number
    if ((size_t)nrec * 64 < SPAN && (MODE == 4 || MODE == 5)) { printf("%s: buffer too small\n", name); return; }
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, buf, nrec, sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * ROUNDS * 8;
    printf("%-9s %.3f ms, %.1f CU-cycles per wave-instruction (%.1f M instructions, %d CUs at %.0f MHz)\n", name, best,
           best * 1e-3 * mhz * 1e6 * cus / instr, instr / 1e6, cus, mhz);
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    const uint32_t nrec = 256u << 20; // 16 GiB of 64-byte records: room for the streaming modes
    char* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, (size_t)nrec * 64));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, (size_t)nrec * 64));
    CHECK(hipMemset(sink, 0, 64));
    run<0>("gather64", buf, 12700000u, sink, cus, mhz);
    run<7>("half64", buf, 12700000u, sink, cus, mhz);
    run<8>("quarter64", buf, 12700000u, sink, cus, mhz);
    run<1>("far64", buf, 12700000u, sink, cus, mhz);
    run<6>("l1set64", buf, 12700000u, sink, cus, mhz);
    run<3>("gather8", buf, 12700000u, sink, cus, mhz);
    run<9>("gather8x4", buf, 12700000u, sink, cus, mhz);
    run<4>("stream8", buf, nrec, sink, cus, mhz);
    run<5>("stream16", buf, nrec, sink, cus, mhz);
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
