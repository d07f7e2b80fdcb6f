// tools/buffer_store_hazard_probe.hip -- the 128-bit buffer-store data hazard found in round 5 (profiles/r05_chain_jit.txt), in isolation.
//
// Every lane stores a float4 of a GOOD value with `buffer_store_dwordx4 v[10:13], voff, rsrc, SOFF offen` and the very next instruction
// (after NOPS wait states) overwrites v10 with a BAD value.  The store must still write GOOD.  Forms:
//     soffset = SGPR (what the compiler emits for a constant offset > 64 passed as soffset; its hazard rule exempts this form)
//     soffset = 0, the offset in the instruction's immediate field (what chain_buffer_store uses since the fix)
// x NOPS = 0, 1 (s_nop 0), 2 (s_nop 1).  The buffer is freshly allocated and never touched before the launch (cold pages: the store's issue
// stalls), 1 wave per workgroup; the host counts floats that came out BAD.
// build: hipcc --offload-arch=gfx950 -O3 tools/buffer_store_hazard_probe.hip -o tools/buffer_store_hazard_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FORM, int NOPS>
__global__ __launch_bounds__(64) void k_store(float* out, int stores_per_wave)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
    const float good = 1.0f, bad = -7.0f;
    // lane's float4 slot: rows of 256 B like the chain kernels' saved activations (64 columns), 32 rows x 2 halves per wave
    unsigned voff = ((blockIdx.x * 32u + (threadIdx.x & 31)) * 64u + 4u * (threadIdx.x >> 5)) * 4u;
    for (int s = 0; s < stores_per_wave; ++s) {
        const unsigned soff = 0xc0;                    // column 48: the offset of the store that was corrupted in the chain kernel
        if constexpr (FORM == 0) {
            asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n\t"
                         ".if %5 == 1\n\ts_nop 0\n\t.endif\n\t.if %5 == 2\n\ts_nop 1\n\t.endif\n\t"
                         "v_mov_b32 v10, %4\n\ts_nop 7"
                         : : "v"(good), "v"(voff), "s"(rs), "s"(soff), "v"(bad), "n"(NOPS) : "v10", "v11", "v12", "v13", "memory");
        } else {
            asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[10:13], %1, %2, 0 offen offset:192\n\t"
                         ".if %4 == 1\n\ts_nop 0\n\t.endif\n\t.if %4 == 2\n\ts_nop 1\n\t.endif\n\t"
                         "v_mov_b32 v10, %3\n\ts_nop 7"
                         : : "v"(good), "v"(voff), "s"(rs), "v"(bad), "n"(NOPS) : "v10", "v11", "v12", "v13", "memory");
        }
        voff += 2048u * 32u * 256u;                   // the next store of this lane: a fresh region (2048 waves x 32 rows x 256 B)
    }
}

template <int FORM, int NOPS>
int run(const char* name)
{
    const int waves = 2048, per = 8;
    const size_t bytes = (size_t)waves * 32 * 256 * per;
    long bad_total = 0, launches = 0;
    for (int rep = 0; rep < 12; ++rep) {
        float* d = nullptr;
        CK(hipMalloc(&d, bytes));                     // fresh, untouched pages every repetition
        hipLaunchKernelGGL((k_store<FORM, NOPS>), dim3(waves), dim3(64), 0, 0, d, per);
        CK(hipDeviceSynchronize());
        std::vector<float> h(bytes / 4);
        CK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
        CK(hipFree(d));
        long bad = 0;
        for (size_t r = 0; r < bytes / 256; ++r)      // the written float4s: columns 48..51 and 52..55 of every 64-float row
            for (int c = 48; c < 56; ++c)
                if (h[r * 64 + c] != 1.0f) ++bad;
        bad_total += bad;
        ++launches;
    }
    printf("%-44s  wrong floats %8ld of %ld (%d launches)\n", name, bad_total, launches * (long)(bytes / 256) * 8, (int)launches);
    return 0;
}

int main()
{
    run<0, 0>("soffset = SGPR,      next instruction writes v10");
    run<0, 1>("soffset = SGPR,      s_nop 0 in between");
    run<0, 2>("soffset = SGPR,      s_nop 1 in between");
    run<1, 0>("immediate offset,    next instruction writes v10");
    run<1, 1>("immediate offset,    s_nop 0 in between");
    run<1, 2>("immediate offset,    s_nop 1 in between (what the compiler inserts)");
    return 0;
}
