// Read-only HBM streaming ceiling on MI355X for k-means-like access: 1.43 GB per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

// variant A: grid-stride, every thread keeps U independent 16-byte loads in flight, no LDS, no barrier
template <int U>
__global__ __launch_bounds__(256) void stream_a(const v4i *__restrict__ src, size_t n16, int *out) {
    v4i acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + 256 * u < n16) ? src[i + 256 * u] : v4i{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
}

// variant B: the k-means structure: 36 KB tile per workgroup iteration, prefetch 1 tile in registers,
// LDS write + 2 barriers per tile, no compute
__global__ __launch_bounds__(256, 3) void stream_b(const v4i *__restrict__ src, int ntiles, int parts, int *out) {
    __shared__ v4i lds[2304 + 128];
    const int tid = threadIdx.x;
    v4i st[9];
    int acc = 0;
    int tile = blockIdx.x;
    if (tile < ntiles)
#pragma unroll
        for (int i = 0; i < 9; ++i) st[i] = src[(size_t)tile * 2304 + tid + 256 * i];
    for (; tile < ntiles; tile += parts) {
#pragma unroll
        for (int i = 0; i < 9; ++i) lds[tid + 256 * i + (i >> 2)] = st[i];
        __syncthreads();
        if (tile + parts < ntiles)
#pragma unroll
            for (int i = 0; i < 9; ++i) st[i] = src[(size_t)(tile + parts) * 2304 + tid + 256 * i];
        acc ^= lds[(tid * 7) % 2304][0];
        __syncthreads();
    }
    if (acc == 0x12345678) out[0] = 1;
}

// variant C: variant B with the real kernel's mapping: grid (parts, images); workgroup (part, b) walks
// image b's tiles part, part+parts, ...  -> 64 concurrent streams 22 MB apart
__global__ __launch_bounds__(256, 3) void stream_c(const v4i *__restrict__ src, int tiles_per_image, int parts, int *out) {
    __shared__ v4i lds[2304 + 128];
    const int tid = threadIdx.x;
    const v4i *base = src + (size_t)blockIdx.y * tiles_per_image * 2304;
    v4i st[9];
    int acc = 0;
    int tile = blockIdx.x;
    if (tile < tiles_per_image)
#pragma unroll
        for (int i = 0; i < 9; ++i) st[i] = base[(size_t)tile * 2304 + tid + 256 * i];
    for (; tile < tiles_per_image; tile += parts) {
#pragma unroll
        for (int i = 0; i < 9; ++i) lds[tid + 256 * i + (i >> 2)] = st[i];
        __syncthreads();
        if (tile + parts < tiles_per_image)
#pragma unroll
            for (int i = 0; i < 9; ++i) st[i] = base[(size_t)(tile + parts) * 2304 + tid + 256 * i];
        acc ^= lds[(tid * 7) % 2304][0];
        __syncthreads();
    }
    if (acc == 0x12345678) out[0] = 1;
}

// variant E (round 2): variant C with the pyramid slab's tile: CH 16-byte chunks per tile (1440 = 23 040 B for the 4x6 bank),
// NST = ceil(CH / 256) clamped loads per thread, WGS workgroups per CU; same LDS write + 2 barriers, no compute.
template <int CH, int WGS>
__global__ __launch_bounds__(256, WGS) void stream_e(const v4i *__restrict__ src, int tiles_per_image, int parts, int *out) {
    constexpr int NST = (CH + 255) / 256;
    __shared__ v4i lds[CH + 128];
    const int tid = threadIdx.x;
    const v4i *base = src + (size_t)blockIdx.y * tiles_per_image * CH;
    v4i st[NST];
    int acc = 0;
    int tile = blockIdx.x;
    if (tile < tiles_per_image)
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = base[(size_t)tile * CH + min(tid + 256 * i, CH - 1)];
    for (; tile < tiles_per_image; tile += parts) {
#pragma unroll
        for (int i = 0; i < NST; ++i) lds[min(tid + 256 * i, CH - 1) + (i >> 2)] = st[i];
        __syncthreads();
        if (tile + parts < tiles_per_image)
#pragma unroll
            for (int i = 0; i < NST; ++i) st[i] = base[(size_t)(tile + parts) * CH + min(tid + 256 * i, CH - 1)];
        acc ^= lds[(tid * 7) % CH][0];
        __syncthreads();
    }
    if (acc == 0x12345678) out[0] = 1;
}

// variant D: LDS-DMA (global_load_lds dwordx4) double/triple buffering: NBUF 36 KB buffers per workgroup,
// NBUF-1 tiles in flight while the current one is "computed"; no VGPR staging, one barrier pair per tile.
template <int NBUF, int WGS>
__global__ __launch_bounds__(256, WGS) void stream_d(const v4i *__restrict__ src, int ntiles, int parts, int *out) {
    extern __shared__ v4i dlds[];                       // NBUF * 2304 chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto dma = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const v4i *g = src + (size_t)tile * 2304 + 256 * i + tid;
            v4i *l = dlds + buf * 2304 + 256 * i + 64 * wave;        // wave-uniform base; lane*16 implied
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)l, 16, 0, 0);
        }
    };
    int acc = 0;
    int tile = blockIdx.x;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (tile + d * parts < ntiles) dma(tile + d * parts, d);
    for (int it = 0; tile < ntiles; tile += parts, ++it) {
        // wait for the oldest tile only: each wave has 9 DMA instructions per tile in flight
        if (NBUF >= 4 && tile + 2 * parts < ntiles) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (NBUF >= 3 && tile + parts < ntiles) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int buf = it % NBUF;
        if (tile + (NBUF - 1) * parts < ntiles) dma(tile + (NBUF - 1) * parts, (it + NBUF - 1) % NBUF);
        acc ^= dlds[buf * 2304 + (tid * 7) % 2304][0];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678) out[0] = 1;
}

// variant F (round 6): the NARROW tile of the split slab (12-bit base: NI items of 16 low bytes + 8 bytes of nibbles, the low bytes
// of the whole tile first, then its nibbles): NR = ceil(NI / 256) rounds of one 16-byte and one 8-byte load per thread (clamped
// to the last item), the LDS image at full width (32 bytes per item), same 2 barriers per tile, no compute.
typedef int v2i __attribute__((ext_vector_type(2)));
template <int NI, int WGS, int STRIDE = NI * 24>
__global__ __launch_bounds__(256, WGS) void stream_f(const unsigned char *__restrict__ src, int ntiles, int parts, int *out) {
    constexpr int NR = (NI + 255) / 256;
    __shared__ v4i lds[2 * NI + 128];
    const int tid = threadIdx.x;
    v4i lo[NR]; v2i mid[NR];
    int acc = 0;
    int tile = blockIdx.x;
    auto load = [&](int t) {
        const unsigned char *tb = src + (size_t)t * STRIDE;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int it = min(tid + 256 * i, NI - 1);
            lo[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(tb) + it);
            mid[i] = __builtin_nontemporal_load(reinterpret_cast<const v2i *>(tb + NI * 16) + it);
        }
    };
    if (tile < ntiles) load(tile);
    for (; tile < ntiles; tile += parts) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int it = min(tid + 256 * i, NI - 1);
            lds[2 * it + (i >> 1)] = lo[i];
            lds[2 * it + 1 + (i >> 1)] = v4i{mid[i][0], mid[i][1], mid[i][0], mid[i][1]};
        }
        __syncthreads();
        if (tile + parts < ntiles) load(tile + parts);
        acc ^= lds[(tid * 7) % (2 * NI)][0];
        __syncthreads();
    }
    if (acc == 0x12345678) out[0] = 1;
}

int main() {
    const size_t bytes = (size_t)64 * 612 * 36864;        // 64 images x 612 tiles x 36 KB (all variants stay inside)
    const size_t n16 = bytes / 16;
    v4i *src; int *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4);
    hipMemset(src, 1, bytes);
    if (getenv("RANDOM_FILL")) {   // same traffic, random payload (HBM / fabric power is data dependent)
        unsigned *h = (unsigned *)malloc(bytes);
        unsigned x = 12345u;
        for (size_t i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
        hipMemcpy(src, h, bytes, hipMemcpyHostToDevice);
        free(h);
        printf("random fill\n");
    }
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    auto time = [&](auto launch, const char *name) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 10; ++r) { hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms; }
        printf("%-40s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
    };
    for (int g : {768, 1536, 2048, 4096})
        { char nm[64]; snprintf(nm, 64, "A: grid-stride U=4 grid=%d", g); time([&] { stream_a<4><<<g, 256>>>(src, n16, out); }, nm); }
    for (int g : {768, 1536, 2048, 4096})
        { char nm[64]; snprintf(nm, 64, "A: grid-stride U=9 grid=%d", g); time([&] { stream_a<9><<<g, 256>>>(src, n16, out); }, nm); }
    const int ntiles = (int)(bytes / 36864);
    for (int g : {512, 768, 1536})
        { char nm[64]; snprintf(nm, 64, "B: kmeans-like tiles grid=%d", g); time([&] { stream_b<<<g, 256>>>(src, ntiles, g, out); }, nm); }
    {
        hipFuncSetAttribute((const void *)stream_d<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 36864);
        hipFuncSetAttribute((const void *)stream_d<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 36864);
        hipFuncSetAttribute((const void *)stream_d<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 36864);
        for (int g : {512, 1024})
            { char nm[64]; snprintf(nm, 64, "D: LDS-DMA 2 buffers grid=%d", g); time([&] { stream_d<2, 2><<<g, 256, 2 * 36864>>>(src, ntiles, g, out); }, nm); }
        for (int g : {256, 512})
            { char nm[64]; snprintf(nm, 64, "D: LDS-DMA 3 buffers grid=%d", g); time([&] { stream_d<3, 1><<<g, 256, 3 * 36864>>>(src, ntiles, g, out); }, nm); }
        for (int g : {256, 512})
            { char nm[64]; snprintf(nm, 64, "D: LDS-DMA 4 buffers grid=%d", g); time([&] { stream_d<4, 1><<<g, 256, 4 * 36864>>>(src, ntiles, g, out); }, nm); }
    }
    {
        const int tpi = 612;   // tiles per 321x488 image
        if ((size_t)64 * tpi * 36864 > bytes) { printf("buffer too small\n"); return 1; }
        for (int parts : {12, 24}) {
            char nm[64]; snprintf(nm, 64, "C: per-image streams 64 x parts=%d", parts);
            time([&] { stream_c<<<dim3(parts, 64), 256>>>(src, tpi, parts, out); }, nm);
        }
    }
    {
        const int tpi = 626;   // tiles per 321x481 image in the round-2 slab (2 501 blocks of 8x8 / 4)
        const size_t b2 = (size_t)64 * tpi * 23040;
        auto time2 = [&](auto launch, const char *name) {
            launch(); hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 10; ++r) { hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms; }
            printf("%-48s %.3f ms  %.0f GB/s\n", name, best, b2 / best / 1e6);
        };
        time2([&] { stream_e<1440, 3><<<dim3(12, 64), 256>>>(src, tpi, 12, out); }, "E: 23 KB tiles, 3 WG/CU, 64 x parts=12");
        time2([&] { stream_e<1440, 4><<<dim3(16, 64), 256>>>(src, tpi, 16, out); }, "E: 23 KB tiles, 4 WG/CU, 64 x parts=16");
        time2([&] { stream_e<1440, 6><<<dim3(24, 64), 256>>>(src, tpi, 24, out); }, "E: 23 KB tiles, 6 WG/CU, 64 x parts=24");
        time2([&] { stream_e<1440, 2><<<dim3(8, 64), 256>>>(src, tpi, 8, out); }, "E: 23 KB tiles, 2 WG/CU, 64 x parts=8");
        // the batch as ONE tile list (what the global-codebook pass does since round 2): all workgroups interleaved
        time2([&] { stream_e<1440, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 23 KB tiles, 3 WG/CU, one list of 768");
        time2([&] { stream_e<1440, 4><<<dim3(1024, 1), 256>>>(src, 64 * tpi, 1024, out); }, "E: 23 KB tiles, 4 WG/CU, one list of 1024");
        time2([&] { stream_e<1440, 2><<<dim3(512, 1), 256>>>(src, 64 * tpi, 512, out); }, "E: 23 KB tiles, 2 WG/CU, one list of 512");
    }
    {
        // round 5: the deep-bank tile (8x8 bank, four levels: 2040 chunks = 32 640 B; 607 tiles per 321x481 image), the same tile
        // padded to 32 768 B, and the 23 KB tile again - constant fill and (RANDOM_FILL=1) random payload
        const int tpi = 607;
        for (int ch : {2040, 2048}) {
            const size_t b2 = (size_t)64 * tpi * ch * 16;
            if (b2 > bytes) { printf("buffer too small\n"); return 1; }
            auto time3 = [&](auto launch, const char *name) {
                launch(); hipDeviceSynchronize();
                float best = 1e9, sum = 0;
                for (int r = 0; r < 10; ++r) { hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms; sum += ms; }
                printf("%-56s best %.3f ms %.0f GB/s | mean %.3f ms %.0f GB/s\n", name, best, b2 / best / 1e6, sum / 10, b2 / (sum / 10) / 1e6);
            };
            if (ch == 2040) {
                time3([&] { stream_e<2040, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 32 640 B tiles, 3 WG/CU, one list of 768");
                time3([&] { stream_e<2040, 2><<<dim3(512, 1), 256>>>(src, 64 * tpi, 512, out); }, "E: 32 640 B tiles, 2 WG/CU, one list of 512");
                time3([&] { stream_e<2040, 4><<<dim3(1024, 1), 256>>>(src, 64 * tpi, 1024, out); }, "E: 32 640 B tiles, 4 WG/CU, one list of 1024");
            } else {
                time3([&] { stream_e<2048, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 32 768 B tiles, 3 WG/CU, one list of 768");
                time3([&] { stream_e<2048, 4><<<dim3(1024, 1), 256>>>(src, 64 * tpi, 1024, out); }, "E: 32 768 B tiles, 4 WG/CU, one list of 1024");
            }
        }
    }
    {
        // round 6: the 12-bit base tiles of the split slab: 17 280 B (4x6 bank; 720 items) and 24 480 B (8x8 bank; 1 020 items),
        // 607 tiles per 321x481 image, one list of the whole batch; E = plain 16-byte chunks, F = the real (16 + 8)-byte items
        const int tpi = 607;
        auto time4 = [&](auto launch, const char *name, size_t b2) {
            launch(); hipDeviceSynchronize();
            float best = 1e9, sum = 0;
            for (int r = 0; r < 10; ++r) { hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms; sum += ms; }
            printf("%-56s best %.3f ms %.0f GB/s | mean %.3f ms %.0f GB/s\n", name, best, b2 / best / 1e6, sum / 10, b2 / (sum / 10) / 1e6);
        };
        const size_t bn = (size_t)64 * tpi * 17280, bd = (size_t)64 * tpi * 24480;
        time4([&] { stream_e<1080, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 17 280 B tiles, 3 WG/CU, one list of 768", bn);
        time4([&] { stream_e<1080, 4><<<dim3(1024, 1), 256>>>(src, 64 * tpi, 1024, out); }, "E: 17 280 B tiles, 4 WG/CU, one list of 1024", bn);
        time4([&] { stream_f<720, 3><<<dim3(768, 1), 256>>>((const unsigned char *)src, 64 * tpi, 768, out); }, "F: 17 280 B (16+8) items, 3 WG/CU, list of 768", bn);
        time4([&] { stream_f<720, 4><<<dim3(1024, 1), 256>>>((const unsigned char *)src, 64 * tpi, 1024, out); }, "F: 17 280 B (16+8) items, 4 WG/CU, list of 1024", bn);
        time4([&] { stream_f<720, 5><<<dim3(1280, 1), 256>>>((const unsigned char *)src, 64 * tpi, 1280, out); }, "F: 17 280 B (16+8) items, 5 WG/CU, list of 1280", bn);
        time4([&] { stream_e<1530, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 24 480 B tiles, 3 WG/CU, one list of 768", bd);
        time4([&] { stream_f<1020, 3><<<dim3(768, 1), 256>>>((const unsigned char *)src, 64 * tpi, 768, out); }, "F: 24 480 B (16+8) items, 3 WG/CU, list of 768", bd);
        time4([&] { stream_f<1020, 4><<<dim3(1024, 1), 256>>>((const unsigned char *)src, 64 * tpi, 1024, out); }, "F: 24 480 B (16+8) items, 4 WG/CU, list of 1024", bd);
        // the same base tiles INSIDE 23 040-byte tiles (the 5 760 bytes of top nibbles behind every base tile skipped): does HBM
        // serve 17 280 of every 23 040 bytes as fast as one contiguous run?
        time4([&] { stream_f<720, 3, 23040><<<dim3(768, 1), 256>>>((const unsigned char *)src, 64 * tpi, 768, out); }, "G: 17 280 of every 23 040 B, 3 WG/CU, list of 768", bn);
        time4([&] { stream_f<720, 4, 23040><<<dim3(1024, 1), 256>>>((const unsigned char *)src, 64 * tpi, 1024, out); }, "G: 17 280 of every 23 040 B, 4 WG/CU, list of 1024", bn);
        time4([&] { stream_f<1020, 3, 32640><<<dim3(768, 1), 256>>>((const unsigned char *)src, 64 * tpi, 768, out); }, "G: 24 480 of every 32 640 B, 3 WG/CU, list of 768", bd);
        time4([&] { stream_e<1440, 3><<<dim3(768, 1), 256>>>(src, 64 * tpi, 768, out); }, "E: 23 040 B tiles (today), 3 WG/CU, list of 768", (size_t)64 * tpi * 23040);
    }
    return 0;
}
