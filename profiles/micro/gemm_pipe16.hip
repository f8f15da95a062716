// Micro-benchmark (not shipped): the software-pipelined fp32 MFMA GEMM of gemm_pipe.hip with the K stage as a template
// parameter: BK = 16 halves the LDS per workgroup (41 KB at 128x128: THREE co-resident workgroups per CU instead of two).
//   C[M,N] = A[M,K] * B[N,K]^T,  both operands K-contiguous (the NT forward shape of GcnSAGELayer).
// What it tries (after reading the schedule of the vendor kernel that reaches 120-142 TF on these shapes):
//   * buffer-descriptor loads: rows past the tile edge and K positions past the row end read as 0 from the
//     hardware range check, so the K loop has no branches and no per-chunk address clamps;
//   * LDS double buffer, ONE barrier per stage; tile t+1 goes registers -> LDS while tile t is multiplied, and
//     the registers are refilled with tile t+2 right after each ds_write frees them;
//   * every memory instruction is placed behind an MFMA (sched_group_barrier), 1 wave per SIMD is enough.
// build: hipcc -O3 --offload-arch=gfx950 profiles/micro/gemm_pipe.hip -o profiles/micro/gemm_pipe
// run:   profiles/micro/gemm_pipe [M N K]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr unsigned SRD_FLAGS = 0x00020000u;

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int M_MFMA = 0x008, M_VMEM_R = 0x020, M_DS_R = 0x100, M_DS_W = 0x200, M_VALU = 0x002, M_SALU = 0x004;

template <int BM, int BN, int WM, int WN, int BK, int EPI>
__global__ void __launch_bounds__(256)
gemm_nt_pipe(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
             int M, int N, int K) {
    constexpr int KPAD = BK + 4, KQ = BK / 4, RP = 256 / KQ;      // k-quads per row, rows per staging pass
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NCA = BM * (BK / 4) / 256, NCB = BN * (BK / 4) / 256;
    constexpr int BUF = (BM + BN) * KPAD;                 // floats per LDS buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // operand windows: rows [m0, m0+rows) x all K; anything outside reads as zero
    const int rowsA = min(BM, M - m0), rowsB = min(BN, N - n0);
    const float* baseA = A + (size_t)m0 * lda;
    const float* baseB = B + (size_t)n0 * ldb;
    const int bytesA = ((rowsA - 1) * lda + K) * 4, bytesB = ((rowsB - 1) * ldb + K) * 4;

    const int kq = tid % KQ, r0 = tid / KQ;
    int voffA[NCA], voffB[NCB];
#pragma unroll
    for (int i = 0; i < NCA; ++i) voffA[i] = ((r0 + RP * i) * lda + kq * 4) * 4;
#pragma unroll
    for (int i = 0; i < NCB; ++i) voffB[i] = ((r0 + RP * i) * ldb + kq * 4) * 4;
    const int wofs = r0 * KPAD + kq * 4;                  // LDS write position of chunk 0 (floats); chunk i: + 32*i*KPAD

    f32x4 ra[NCA], rb[NCB];
    int vc_ld = 0, vc_st = 0;                             // valid k count of this lane's chunk (<=0: none, >=4: all)

    auto issue_loads = [&](int t, auto FROM, auto TO) {   // chunks [FROM, TO) of tile t (A chunks first, then B)
        const int kl = t * BK;
        const int remA = bytesA - kl * 4, remB = bytesB - kl * 4;
        __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc((void*)(baseA + kl), 0, remA > 0 ? remA : 0, SRD_FLAGS);
        __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc((void*)(baseB + kl), 0, remB > 0 ? remB : 0, SRD_FLAGS);
        // chunks that start at or past K: push the offset out of the window (plain arithmetic: a ?: here became a branch)
        const int oob = ((K - kl - kq * 4 - 1) >> 31) & (int)0x80000000;
#pragma unroll
        for (int j = decltype(FROM)::value; j < decltype(TO)::value; ++j) {
            if (j < NCA) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa, voffA[j] | oob, 0, 0));
            else rb[j - NCA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sb, voffB[j - NCA] | oob, 0, 0));
        }
    };
    auto store_chunks = [&](int buf, auto FROM, auto TO) {
        float* wa = lds + buf * BUF + wofs;
        float* wb = wa + BM * KPAD;
#pragma unroll
        for (int j = decltype(FROM)::value; j < decltype(TO)::value; ++j) {
            f32x4 v = j < NCA ? ra[j < NCA ? j : 0] : rb[j >= NCA ? j - NCA : 0];
            v.y = vc_st > 1 ? v.y : 0.f;
            v.z = vc_st > 2 ? v.z : 0.f;
            v.w = vc_st > 3 ? v.w : 0.f;
            if (j < NCA) *reinterpret_cast<f32x4*>(wa + j * RP * KPAD) = v;
            else *reinterpret_cast<f32x4*>(wb + (j - NCA) * RP * KPAD) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 fa[2][TM], fb[2][TN];
    const int frA = ((wm * TM) * 32 + (lane & 31)) * KPAD + (lane >> 5) * 4;
    const int frB = BM * KPAD + ((wn * TN) * 32 + (lane & 31)) * KPAD + (lane >> 5) * 4;
    auto read_frags = [&](int buf, int kg, int slot) {
        const float* pa = lds + buf * BUF + frA + kg * 8;
        const float* pb = lds + buf * BUF + frB + kg * 8;
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[slot][a] = *reinterpret_cast<const f32x4*>(pa + a * 32 * KPAD);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[slot][b] = *reinterpret_cast<const f32x4*>(pb + b * 32 * KPAD);
    };
    auto mfma_group = [&](int slot) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][tt], fb[slot][b][tt], acc[a][b], 0, 0, 0);
    };
    constexpr int NC = NCA + NCB, H = NC / 2;
    constexpr int NMF = 4 * TM * TN;                       // MFMAs per k-group
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, H>;
    using IN = std::integral_constant<int, NC>;

    const int T = (K + BK - 1) / BK;
    // prologue: tile 0 -> LDS[0], tile 1 -> registers
    issue_loads(0, I0{}, IN{});
    vc_st = K - kq * 4;
    store_chunks(0, I0{}, IN{});
    issue_loads(1, I0{}, IN{});
    vc_st = K - BK - kq * 4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_frags(0, 0, 0);

    if constexpr (BK == 32) {
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        // k-group 0: fragments of k-group 1; first half of tile t+1 -> LDS[cur^1], refill with tile t+2
        read_frags(cur, 1, 1);
        store_chunks(cur ^ 1, I0{}, IH{});
        issue_loads(t + 2, I0{}, IH{});
        mfma_group(0);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
        for (int i = 0; i < H; ++i) { SGB(M_MFMA, 1); SGB(M_DS_W, 1); SGB(M_MFMA, 1); SGB(M_VMEM_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN) - 2 * H);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 1
        read_frags(cur, 2, 0);
        store_chunks(cur ^ 1, IH{}, IN{});
        issue_loads(t + 2, IH{}, IN{});
        mfma_group(1);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
        for (int i = 0; i < NC - H; ++i) { SGB(M_MFMA, 1); SGB(M_DS_W, 1); SGB(M_MFMA, 1); SGB(M_VMEM_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN) - 2 * (NC - H));
        __builtin_amdgcn_sched_barrier(0);
        vc_st = K - (t + 2) * BK - kq * 4;
        // k-group 2
        read_frags(cur, 3, 1);
        mfma_group(0);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // k-group 3: first fragments of tile t+1
        read_frags(cur ^ 1, 0, 0);
        mfma_group(1);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN));
        __builtin_amdgcn_sched_barrier(0);
    }

    } else {
    // BK = 16: two k-groups per stage.  k-group 0 carries the whole staging traffic of tile t+1 (LDS writes, refills with
    // tile t+2) and the reads of k-group 1's fragments; the barrier; k-group 1 hides the first fragment reads of tile t+1.
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        read_frags(cur, 1, 1);
        store_chunks(cur ^ 1, I0{}, IN{});
        issue_loads(t + 2, I0{}, IN{});
        mfma_group(0);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
        for (int i = 0; i < NC; ++i) { SGB(M_MFMA, 1); SGB(M_DS_W, 1); SGB(M_MFMA, 1); SGB(M_VMEM_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN) - 2 * NC);
        __builtin_amdgcn_sched_barrier(0);
        vc_st = K - (t + 2) * BK - kq * 4;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(cur ^ 1, 0, 0);
        mfma_group(1);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
        SGB(M_MFMA, NMF - (TM + TN));
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    if constexpr (EPI == 2) {                                 // ablation: no C stores (keep the accumulators alive)
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        if (s == 12345.678f) C[0] = s;
        return;
    }
    if constexpr (EPI == 1) {
        // wave-private [TM*32][TN*32 + 4] image in LDS (the operand buffers are dead), then 16-byte row pieces:
        // 4x fewer store instructions, every store instruction writes whole 128-B lines
        constexpr int WR = TM * 32, WC = TN * 32, LDW = WC + 4;
        static_assert(4 * WR * LDW * 4 <= 2 * (BM + BN) * KPAD * 4, "epilogue image fits the operand buffers");
        __syncthreads();                                       // every wave is done reading operand fragments
        float* img = lds + wave * WR * LDW;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    img[(a * 32 + hrow + (r & 3) + 8 * (r >> 2)) * LDW + b * 32 + col_l] = acc[a][b][r];
        // same wave reads back: LDS is in order per wave
        constexpr int LPRW = WC / 4, RPI = 64 / LPRW;          // lanes per row, rows per instruction
        const int rr = lane / LPRW, cc = (lane % LPRW) * 4;
#pragma unroll
        for (int j = 0; j < WR / RPI; ++j) {
            const int lr = j * RPI + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + lr * LDW + cc);
            const int row = m0 + wm * WR + lr, col = n0 + wn * WC + cc;
            if (row < M && col + 3 < N) *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v;
            else if (row < M) { for (int q = 0; q < 4; ++q) if (col + q < N) C[(size_t)row * ldc + col + q] = v[q]; }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wn * TN + b) * 32 + col_l;
        if (col >= N) continue;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int rbase = m0 + (wm * TM + a) * 32 + hrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int WM, int WN, int BK, int EPI = 0>
void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
         const std::vector<float>& hB) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const size_t shm = 2 * (BM + BN) * (BK + 4) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_nt_pipe<BM, BN, WM, WN, BK, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    auto launch = [&]() { hipLaunchKernelGGL((gemm_nt_pipe<BM, BN, WM, WN, BK, EPI>), dim3(tiles), dim3(256), shm, 0, dA, K, dB, K, dC, N, M, N, K); };
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    // integer-valued inputs: every product and partial sum is exact in fp32, so the check is equality
    long bad = 0;
    for (int s = 0; s < 4000; ++s) {
        const int m = (s * 7919 + (s % 3 ? M - 1 - s % 200 : 0)) % M, n = (s * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        if (EPI != 2 && (double)hC[(size_t)m * N + n] != ref) { if (bad < 5) printf("  mismatch (%d,%d): %f vs %f\n", m, n, hC[(size_t)m * N + n], ref); ++bad; }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-22s M=%d N=%d K=%d tiles=%d  %8.1f us  %6.1f TF  %s\n", name, M, N, K, tiles, ms * 1e3, 2.0 * M * N * K / ms / 1e9,
           bad ? "MISMATCH" : "exact");
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24495, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1662;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 28) - 8); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 29) - 4); }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 128, 2, 2, 32, 0>("128x128 dword stores", dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 2, 2, 32, 1>("128x128 LDS->x4 stores", dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 2, 2, 32, 2>("128x128 no stores", dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, 32, 0>("64x128 dword stores", dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, 32, 1>("64x128 LDS->x4 stores", dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, 32, 2>("64x128 no stores", dA, dB, dC, M, N, K, hA, hB);
    }
    return 0;
}
