// Probe (round 4): what does the memory system give a write-only epilogue, by store pattern?
// Every fp16-activation kernel's epilogue stores 16 bytes per lane; a wave instruction covers 8 rows x 128 B (plain fp16 rows, 64-column
// blocks) or 8 rows x 64 B (the GEGLU epilogue: 32 outputs per 64 accumulator columns), rows `ld` bytes apart; the other half of a
// 128-B line is written microseconds later by another column block.  The timing ablations of gemm_f16dma (profiles/r4_gemm_f16dma_ablate.txt)
// put 0.35 ms of the plain and 0.51 ms of the GEGLU 320 -> 2 560 projection into the epilogue: 1.9 / 0.66 TB/s of output.
//   mode 0: fully contiguous 16 B per lane (1 KB per wave instruction)            -- the ceiling
//   mode 1: 8 rows x 128 B per wave instruction, rows ld bytes apart, column blocks swept row-tile by row-tile (the GEMM's order)
//   mode 2: 8 rows x  64 B per wave instruction (half lines), the other half written by a LATER sweep
//   mode 3: as 2, but both halves written back to back by the same wave (what a fused two-block GEGLU pass would do)
//   mode 4: 32 rows x 32 B per wave instruction (lane = row, lane half = 16-B piece): what an epilogue WITHOUT the LDS transpose would
//           store if the MFMA operands are swapped (lane = pixel, registers = channels; v_permlane32_swap pairs the quads) -- and `rd<4>`
//           the residual rows it would load in the same pattern
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/hbm_write_rate tools/probes/hbm_write_rate.hip && tools/probes/hbm_write_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// out: rows x ld bytes.  A workgroup (256 threads = 4 waves) owns a tile of 128 rows x 512 B (4 blocks of 128 B); grid covers the matrix.
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) wr(char* out, long long rows, long long ld, int col_tiles) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long tile = blockIdx.x;
    const long long rt = tile / col_tiles, ct = tile % col_tiles;
    const f4 v = {1.f, 2.f, 3.f, (float)tid};
    auto st = [&](char* p) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = v; };
    if (MODE == 0) {
        char* base = out + (tile * 128 * 512);
#pragma unroll
        for (int i = 0; i < 16; ++i) st(base + ((long long)i * 256 + tid) * 16);
        return;
    }
    char* base = out + (rt * 128 + wave * 32) * ld + ct * 512;          // this wave: 32 rows x 512 B
    if (MODE == 1) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) st(base + (long long)(pass * 8 + (lane >> 3)) * ld + blk * 128 + (lane & 7) * 16);
    } else if (MODE == 2) {                          // half lines: first all left halves of the 4 blocks, then (later) all right halves
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
                    st(base + (long long)(pass * 16 + (lane >> 2)) * ld + blk * 128 + half * 64 + (lane & 3) * 16);
    } else if (MODE == 4) {                          // lane = row (32 rows), lane half = which 16 B of a 32-B piece; 16 pieces per 512-B row
#pragma unroll
        for (int piece = 0; piece < 16; ++piece) st(base + (long long)(lane & 31) * ld + piece * 32 + (lane >> 5) * 16);
    } else {                                         // both halves of a line back to back
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    st(base + (long long)(pass * 16 + (lane >> 2)) * ld + blk * 128 + half * 64 + (lane & 3) * 16);
    }
}

// the same tiles READ (residual rows): mode 1 = 8 rows x 128 B per instruction, mode 4 = 32 rows x 32 B
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) rd(const char* in, long long rows, long long ld, int col_tiles, float* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long tile = blockIdx.x;
    const long long rt = tile / col_tiles, ct = tile % col_tiles;
    const char* base = in + (rt * 128 + wave * 32) * ld + ct * 512;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    auto ldv = [&](const char* p) { return NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)) : *reinterpret_cast<const f4*>(p); };
    if (MODE == 1) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) acc += ldv(base + (long long)(pass * 8 + (lane >> 3)) * ld + blk * 128 + (lane & 7) * 16);
    } else {
#pragma unroll
        for (int piece = 0; piece < 16; ++piece) acc += ldv(base + (long long)(lane & 31) * ld + piece * 32 + (lane >> 5) * 16);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[tid] = acc[0];
}

template <int MODE, bool NT>
void run_rd(const char* label, const char* in, long long rows, long long ld, float* sink) {
    const int col_tiles = (int)(ld / 512);
    const long long tiles = rows / 128 * col_tiles;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd<MODE, NT><<<(unsigned)tiles, 256>>>(in, rows, ld, col_tiles, sink);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        rd<MODE, NT><<<(unsigned)tiles, 256>>>(in, rows, ld, col_tiles, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-72s %8.3f ms  %6.2f TB/s\n", label, best, (double)rows * ld / best / 1e9);
}

template <int MODE, bool NT>
void run(const char* label, char* out, long long rows, long long ld) {
    const int col_tiles = (int)(ld / 512);
    const long long tiles = rows / 128 * col_tiles;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    wr<MODE, NT><<<(unsigned)tiles, 256>>>(out, rows, ld, col_tiles);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        wr<MODE, NT><<<(unsigned)tiles, 256>>>(out, rows, ld, col_tiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-72s %8.3f ms  %6.2f TB/s\n", label, best, (double)rows * ld / best / 1e9);
}

int main() {
    const long long rows = 131072, ld = 5120;          // 131 072 x 2 560 fp16 = 671 MB: the plain 320 -> 2 560 projection's output
    char* out; (void)hipMalloc(&out, rows * ld);
    printf("# write-only store patterns, %lld rows x %lld B = %.0f MB (MI355X)\n", rows, ld, rows * ld / 1e6);
    run<0, false>("contiguous 1 KB per wave instruction", out, rows, ld);
    run<0, true>("contiguous 1 KB per wave instruction, nontemporal", out, rows, ld);
    run<1, false>("8 rows x 128 B per instruction (plain fp16 epilogue)", out, rows, ld);
    run<1, true>("8 rows x 128 B per instruction, nontemporal", out, rows, ld);
    run<2, false>("16 rows x 64 B per instruction, other half later (GEGLU epilogue)", out, rows, ld);
    run<2, true>("16 rows x 64 B per instruction, other half later, nontemporal", out, rows, ld);
    run<3, false>("16 rows x 64 B per instruction, both halves back to back", out, rows, ld);
    run<3, true>("16 rows x 64 B per instruction, both halves back to back, nontemporal", out, rows, ld);
    run<4, false>("32 rows x 32 B per instruction (lane = row; no-transpose epilogue)", out, rows, ld);
    run<4, true>("32 rows x 32 B per instruction, nontemporal", out, rows, ld);
    float* sink; (void)hipMalloc(&sink, 4096);
    run_rd<1, false>("READ  8 rows x 128 B per instruction", out, rows, ld, sink);
    run_rd<1, true>("READ  8 rows x 128 B per instruction, nontemporal", out, rows, ld, sink);
    run_rd<4, false>("READ 32 rows x 32 B per instruction", out, rows, ld, sink);
    run_rd<4, true>("READ 32 rows x 32 B per instruction, nontemporal", out, rows, ld, sink);
    (void)hipFree(sink);
    (void)hipFree(out);
    return 0;
}
