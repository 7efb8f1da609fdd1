// Development aid: issue rate of v_mfma_f64_16x16x4_f64 as a function of where its accumulator lives -- written in
// assembly so that the register assignment is what it says: (a) D = C, vector registers; (b) D != C, vector registers
// (results never read: no dependencies at all); (c) D = C in accumulation registers (AGPRs); (d) D != C, AGPRs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_issue.hip -o mfma_issue && ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define M(D, C) "v_mfma_f64_16x16x4_f64 " D ", v[0:1], v[2:3], " C "\n"

template <int MODE>
__global__ __launch_bounds__(256) void spin(double *out, int iters, const char *gbuf = nullptr)
{
    __shared__ double lds_buf[4096];
    // MODE 11+: this lane's 16 bytes of a 1 KiB piece (an L2-resident source), landing in this wavefront's KiB of LDS
    const char *gsrc = gbuf + (size_t)(blockIdx.x % 256) * 4096 + (threadIdx.x / 64) * 1024 + (threadIdx.x & 63) * 16;
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((unsigned)(size_t)(__attribute__((address_space(3))) double *)lds_buf + 16384 + (threadIdx.x / 64) * 1024));
    lds_buf[threadIdx.x] = 1.0 + threadIdx.x;
    lds_buf[threadIdx.x + 256] = 2.0;
    __syncthreads();
    // MODE 5: lane-contiguous reads; MODE 6: the Gram kernel's fragment pattern (column lane & 15 of a [column][34] tile, row lane >> 4)
    const unsigned lane = threadIdx.x & 63;
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds_buf +
                              (MODE == 6 || MODE >= 14 ? 8 * ((lane & 15) * 34 + (lane >> 4)) : 8 * lane);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile(M("v[8:15]", "v[8:15]") M("v[16:23]", "v[16:23]") M("v[24:31]", "v[24:31]") M("v[32:39]", "v[32:39]")
                         M("v[40:47]", "v[40:47]") M("v[48:55]", "v[48:55]") M("v[56:63]", "v[56:63]") M("v[64:71]", "v[64:71]")
                         ::: "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        if (MODE == 1)
            asm volatile(M("v[8:15]", "v[72:79]") M("v[16:23]", "v[80:87]") M("v[24:31]", "v[88:95]") M("v[32:39]", "v[96:103]")
                         M("v[40:47]", "v[104:111]") M("v[48:55]", "v[112:119]") M("v[56:63]", "v[120:127]") M("v[64:71]", "v[128:135]")
                         ::: "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        if (MODE == 2)
            asm volatile(M("a[8:15]", "a[8:15]") M("a[16:23]", "a[16:23]") M("a[24:31]", "a[24:31]") M("a[32:39]", "a[32:39]")
                         M("a[40:47]", "a[40:47]") M("a[48:55]", "a[48:55]") M("a[56:63]", "a[56:63]") M("a[64:71]", "a[64:71]")
                         ::: "v0", "v1", "v2", "v3", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",
                         "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",
                         "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51",
                         "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67",
                         "a68", "a69", "a70", "a71");
        if (MODE == 3)
            asm volatile(M("a[8:15]", "a[72:79]") M("a[16:23]", "a[80:87]") M("a[24:31]", "a[88:95]") M("a[32:39]", "a[96:103]")
                         M("a[40:47]", "a[104:111]") M("a[48:55]", "a[112:119]") M("a[56:63]", "a[120:127]") M("a[64:71]", "a[128:135]")
                         ::: "v0", "v1", "v2", "v3", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",
                         "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",
                         "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51",
                         "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67",
                         "a68", "a69", "a70", "a71");
        if (MODE == 4)          // A and B in the same register banks (v0:1 and v4:5), D = C
            asm volatile("v_mfma_f64_16x16x4_f64 v[8:15], v[0:1], v[4:5], v[8:15]\n v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[4:5], v[16:23]\n"
                         "v_mfma_f64_16x16x4_f64 v[24:31], v[0:1], v[4:5], v[24:31]\n v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[4:5], v[32:39]\n"
                         "v_mfma_f64_16x16x4_f64 v[40:47], v[0:1], v[4:5], v[40:47]\n v_mfma_f64_16x16x4_f64 v[48:55], v[0:1], v[4:5], v[48:55]\n"
                         "v_mfma_f64_16x16x4_f64 v[56:63], v[0:1], v[4:5], v[56:63]\n v_mfma_f64_16x16x4_f64 v[64:71], v[0:1], v[4:5], v[64:71]\n"
                         ::: "v0", "v1", "v4", "v5", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        if (MODE == 5 || MODE == 6)   // operands refreshed from LDS before every MFMA, the way the Gram kernel does it
            asm volatile("ds_read_b64 v[0:1], %0\n ds_read_b64 v[2:3], %0 offset:512\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[8:15], v[72:73], v[74:75], v[8:15]\n"
                         "ds_read_b64 v[72:73], %0 offset:1024\n ds_read_b64 v[74:75], %0 offset:1536\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[2:3], v[16:23]\n"
                         "ds_read_b64 v[0:1], %0 offset:2048\n ds_read_b64 v[2:3], %0 offset:2560\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[24:31], v[72:73], v[74:75], v[24:31]\n"
                         "ds_read_b64 v[72:73], %0 offset:3072\n ds_read_b64 v[74:75], %0 offset:3584\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[2:3], v[32:39]\n"
                         "ds_read_b64 v[0:1], %0 offset:4096\n ds_read_b64 v[2:3], %0 offset:4608\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[40:47], v[72:73], v[74:75], v[40:47]\n"
                         "ds_read_b64 v[72:73], %0 offset:5120\n ds_read_b64 v[74:75], %0 offset:5632\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[48:55], v[0:1], v[2:3], v[48:55]\n"
                         "ds_read_b64 v[0:1], %0 offset:6144\n ds_read_b64 v[2:3], %0 offset:6656\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[56:63], v[72:73], v[74:75], v[56:63]\n"
                         "ds_read_b64 v[72:73], %0 offset:7168\n ds_read_b64 v[74:75], %0 offset:7680\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 v[64:71], v[0:1], v[2:3], v[64:71]\n"
                         :: "v"(lds_addr) : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
        // MODE 7 / 8 / 9 / 10: D = C in vector registers with other VALU work of the same wavefront between the MFMAs --
        // two 32-bit integer adds, four of them, two fp64 FMAs, four fp64 FMAs per MFMA (all independent of the MFMAs):
        // does the matrix instruction share the SIMD's vector datapath with them or run beside it?
#define CLOB ::: "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
                         "v68", "v69", "v70", "v71", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87"
#define I2 "v_add_u32 v76, v76, v77\n v_add_u32 v78, v78, v77\n"
#define F2 "v_fma_f64 v[80:81], v[80:81], v[82:83], v[82:83]\n v_fma_f64 v[84:85], v[84:85], v[86:87], v[86:87]\n"
#define EIGHT(X) M("v[8:15]", "v[8:15]") X M("v[16:23]", "v[16:23]") X M("v[24:31]", "v[24:31]") X M("v[32:39]", "v[32:39]") X \
                 M("v[40:47]", "v[40:47]") X M("v[48:55]", "v[48:55]") X M("v[56:63]", "v[56:63]") X M("v[64:71]", "v[64:71]") X
        if (MODE == 7) asm volatile(EIGHT(I2) CLOB);
        if (MODE == 8) asm volatile(EIGHT(I2 I2) CLOB);
        if (MODE == 9) asm volatile(EIGHT(F2) CLOB);
        if (MODE == 10) asm volatile(EIGHT(F2 F2) CLOB);
        // MODE 11 / 12 / 13: one / two / four LDS-DMA loads (global_load_lds_dwordx4, 64-bit per-lane addresses, data from L2)
        // per eight MFMAs: what a piece of the Gram kernel's staging costs the matrix stream of its SIMD
#define DMA1 "s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, off\n"
        if (MODE == 11)
            asm volatile(DMA1 M("v[8:15]", "v[8:15]") M("v[16:23]", "v[16:23]") M("v[24:31]", "v[24:31]") M("v[32:39]", "v[32:39]")
                         M("v[40:47]", "v[40:47]") M("v[48:55]", "v[48:55]") M("v[56:63]", "v[56:63]") M("v[64:71]", "v[64:71]")
                         "s_waitcnt vmcnt(0)\n" :: "v"(gsrc), "s"(lds_dst) : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        if (MODE == 12)
            asm volatile(DMA1 M("v[8:15]", "v[8:15]") M("v[16:23]", "v[16:23]") M("v[24:31]", "v[24:31]") M("v[32:39]", "v[32:39]")
                         DMA1 M("v[40:47]", "v[40:47]") M("v[48:55]", "v[48:55]") M("v[56:63]", "v[56:63]") M("v[64:71]", "v[64:71]")
                         "s_waitcnt vmcnt(0)\n" :: "v"(gsrc), "s"(lds_dst) : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        if (MODE == 13)
            asm volatile(DMA1 M("v[8:15]", "v[8:15]") M("v[16:23]", "v[16:23]") DMA1 M("v[24:31]", "v[24:31]") M("v[32:39]", "v[32:39]")
                         DMA1 M("v[40:47]", "v[40:47]") M("v[48:55]", "v[48:55]") DMA1 M("v[56:63]", "v[56:63]") M("v[64:71]", "v[64:71]")
                         "s_waitcnt vmcnt(0)\n" :: "v"(gsrc), "s"(lds_dst) : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71");
        // MODE 14 / 15: the Gram fragment pattern (operands from LDS before every MFMA) WITH two / four LDS-DMA loads per eight
        // MFMAs landing in the same LDS: do the arriving pieces get in the way of the fragment reads?
#define LM(D, A, B, OA, OB) "ds_read_b64 " A ", %2 offset:" OA "\n ds_read_b64 " B ", %2 offset:" OB "\n s_waitcnt lgkmcnt(2)\n v_mfma_f64_16x16x4_f64 " D ", " B ", " A ", " D "\n"
        if (MODE == 14)
            asm volatile(DMA1 LM("v[8:15]", "v[0:1]", "v[2:3]", "0", "512") LM("v[16:23]", "v[72:73]", "v[74:75]", "1024", "1536")
                         LM("v[24:31]", "v[0:1]", "v[2:3]", "2048", "2560") LM("v[32:39]", "v[72:73]", "v[74:75]", "3072", "3584")
                         DMA1 LM("v[40:47]", "v[0:1]", "v[2:3]", "4096", "4608") LM("v[48:55]", "v[72:73]", "v[74:75]", "5120", "5632")
                         LM("v[56:63]", "v[0:1]", "v[2:3]", "6144", "6656") LM("v[64:71]", "v[72:73]", "v[74:75]", "7168", "7680")
                         "s_waitcnt vmcnt(0)\n" :: "v"(gsrc), "s"(lds_dst), "v"(lds_addr) : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                         "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                         "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
                         "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    }
    if (out && iters < 0) out[threadIdx.x] = 1.0;
}

template <int MODE>
void run(const char *what, int waves_per_simd)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    static char *gbuf = nullptr;
    if (!gbuf) (void)hipMalloc(&gbuf, 2 << 20);
    for (int warm = 0; warm < 5; ++warm) spin<MODE><<<blocks, 256>>>(nullptr, iters, gbuf);
    (void)hipEventRecord(e0);
    spin<MODE><<<blocks, 256>>>(nullptr, iters, gbuf);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 8.0 * iters * waves_per_simd;
    printf("%-34s %d wave(s) per SIMD: %6.1f TFLOP/s, %5.1f cycles per instruction and SIMD at 2.39 GHz\n", what, waves_per_simd,
           2048.0 * 8.0 * iters * 4.0 * blocks / ms / 1e9, ms * 1e-3 * 2.39e9 / instr_per_simd);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("D = C, vector registers:", w);
        run<1>("D != C, vector registers:", w);
        run<2>("D = C, accumulation registers:", w);
        run<3>("D != C, accumulation registers:", w);
        run<4>("A, B in the same banks, D = C:", w);
        run<5>("operands from LDS every time:", w);
        run<6>("... in the Gram fragment pattern:", w);
        run<7>("+ 2 integer adds per MFMA:", w);
        run<8>("+ 4 integer adds per MFMA:", w);
        run<9>("+ 2 fp64 FMAs per MFMA:", w);
        run<10>("+ 4 fp64 FMAs per MFMA:", w);
        run<11>("+ 1 LDS-DMA load per 8 MFMAs:", w);
        run<12>("+ 2 LDS-DMA loads per 8 MFMAs:", w);
        run<13>("+ 4 LDS-DMA loads per 8 MFMAs:", w);
        run<14>("LDS operands + 2 LDS-DMA per 8:", w);
    }
    return 0;
}
