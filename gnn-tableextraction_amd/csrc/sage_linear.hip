// Per-node transform of GcnSAGELayer on gfx950: fp32 MFMA GEMMs + LayerNorm/ReLU.
//
// replaces (reference src/components/graphs/models.py):
//   :69-72  torch.cat((h, ah*norm), 1)      -> never materialised: the A operand is read from two
//                                              buffers (K segments [0,k1) and [k1,k1+k2))
//   :63     nn.Linear(2F, out)              -> v_mfma_f32_32x32x2_f32 GEMM, bias in the epilogue
//   :64-66  nn.LayerNorm(out) + activation  -> row-wise kernel (one wave per node row)
// and their autograd: dX = dZ W (NN), dW = dZ^T X (TN, split over the node dimension with a
// deterministic slab reduction), LayerNorm/ReLU backward with column sums for dgamma/dbeta/dbias.
//
// Roofline: MFMA (fp32 matrix peak 157.3 TFLOP/s; v_mfma_f32_32x32x2_f32 is bit-exact fp32 FMA
// chains, no TF32-like shortcut exists on gfx950).  LayerNorm passes are HBM-bound.
//
// GEMM structure (one workgroup = 4 waves, wave64):
//   block tile BM x BN (128x128, 128x32 or 32x128), BK = 32; every wave owns a (BM/WM)x(BN/WN) patch
//   of 32x32 MFMA tiles held in accumulator registers.
//   global -> registers (dwordx4, prefetch of tile t+1 issued before the MFMAs of tile t)
//          -> LDS (ds_write_b128) -> fragments.
//   K-contiguous operands ([rows][K] in memory) keep [rows][BK+4] LDS images: the +4 pad makes the
//   16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots; one b128 read feeds 4 MFMAs
//   (k = k0+4h+t, h = lane>>5: any k order is valid as long as A and B agree).
//   Row-contiguous operands ([K][rows]) keep [BK][rows+4] images read with conflict-free ds_read_b32.
//   blockIdx is remapped so that the tiles sharing an A row-panel run on one XCD (its L2 keeps it).
#include "gte_common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N).  The staging registers are
// arrays indexed by chunk number; a runtime index (even one that unrolls to a constant later) lets
// hipcc demote them to scratch/LDS ("promote alloca"), so every chunk index is a template constant.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

constexpr int BK = 32;
constexpr int KPAD = BK + 4;      // row stride (floats) of a K-contiguous LDS image

struct GemmParams {
    // A(m,k): segment 0 = A1 for k in [0,K1), segment 1 = A2 for k in [K1, K1+K2)
    const float* A1; int64_t lda1; int K1;
    const float* A2; int64_t lda2; int K2;
    const float* B; int64_t ldb;       // B(k,n); k runs over K1+K2 (segment 1 starts at row/col K1)
    const float* Bn2; int64_t ldbn2;   // optional second B for output columns [Nseg, 2*Nseg): C = A [B | Bn2]
    int Nseg;                          // columns per N segment when Bn2 != nullptr (then N == 2*Nseg)
    float* C; int64_t ldc;
    const float* bias;                 // nullable, per column n
    int M, N;
    int relu;                          // epilogue max(0, .) (only without split-K)
    int accumulate;                    // C += (only without split-K)
    int splits;                        // >1: write partial slabs to `slab` instead of C
    float* slab;                       // [splits][M][N] (ld = N)
    int tiles_per_split;               // K tiles (of BK) per split
};

// ---- staging: one BMxBK (or BKxBM) operand tile, global -> regs -> LDS ---------------------------
// Everything that does not depend on the K position is computed ONCE per K segment (init): the
// per-chunk row pointers (clamped into the matrix, so every load is in bounds and unconditional) and,
// for row-contiguous operands, the shift of a chunk that straddles the last row.  Inside the K loop an
// interior stage is just `pointer + k offset -> global_load_dwordx4 -> ds_write_b128`; only the last
// K tile of a segment (K % 32 != 0) and the edge M/N tiles take the guarded path, selected by a
// block-uniform flag.  (Measured on the first version, which recomputed addresses and guards per
// chunk per stage: ~300 VALU instructions per stage per wave beside 32 MFMAs kept the MFMA pipe at
// 62 % busy; and a version with `if (in range) load` serialised the loads -- one exec-masked branch
// with s_waitcnt vmcnt(0) per chunk.)
// Rows past the end of the matrix are NOT zeroed: they only feed accumulator rows/columns that the
// epilogue never stores.  K positions past the end ARE zeroed (they feed every output).
//   meta = s in 0..3 : element j of the chunk is loaded[j + s] (zero if j + s > 3);  meta >= 4 : zeros.
// Requires every contiguous run to hold >= 4 floats (the host routes smaller shapes to gemm_small_kernel).
__device__ __forceinline__ float4 shift_chunk(const f4u t, int s) {
    float4 v;
    v.x = s == 0 ? t.x : s == 1 ? t.y : s == 2 ? t.z : s == 3 ? t.w : 0.f;
    v.y = s == 0 ? t.y : s == 1 ? t.z : s == 2 ? t.w : 0.f;
    v.z = s == 0 ? t.z : s == 1 ? t.w : 0.f;
    v.w = s == 0 ? t.w : 0.f;
    return v;
}

// K-contiguous source ([rows][K], ld): chunk c -> (row = c / 8, kq = c % 8), 4 consecutive k.
template <int ROWS>
struct StageK {
    static constexpr int CHUNKS = ROWS * (BK / 4);
    static constexpr int PER_THREAD = (CHUNKS + 255) / 256;
    const float* base[PER_THREAD];      // &src[row][kq*4], row clamped
    f4u r[PER_THREAD];
    int meta[PER_THREAD];
    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = min(tid + i * 256, CHUNKS - 1);
            base[i] = src + (int64_t)min(row0 + c / (BK / 4), nrows - 1) * ld + (c % (BK / 4)) * 4;
        }
    }
    __device__ __forceinline__ bool row_edge() const { return false; }
    // one chunk of the tile starting at k = kl of a segment of length kseg; k_edge = (kl + BK > kseg), uniform
    template <int i>
    __device__ __forceinline__ void load_chunk(int kl, int kseg, bool k_edge, int tid) {
        if (!k_edge) {
            r[i] = *reinterpret_cast<const f4u*>(base[i] + kl);
        } else {
            const int c = min(tid + i * 256, CHUNKS - 1);
            const int k = kl + (c % (BK / 4)) * 4;
            const int kk = min(k, kseg - 4);
            meta[i] = k < kseg ? k - kk : 4;
            r[i] = *reinterpret_cast<const f4u*>(base[i] + kl + (kk - k));
        }
    }
    template <int i>
    __device__ __forceinline__ void store_chunk(float* __restrict__ lds, bool edge, int tid) const {
        const int c = tid + i * 256;
        if (CHUNKS % 256 == 0 || c < CHUNKS) {
            float4 v = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
            if (edge) v = shift_chunk(r[i], meta[i]);
            *reinterpret_cast<float4*>(lds + (c / (BK / 4)) * KPAD + (c % (BK / 4)) * 4) = v;
        }
    }
};

// Row-contiguous source ([K][rows], ld): chunk c -> (k = c / (ROWS/4), rq = c % (ROWS/4)), 4 consecutive rows.
template <int ROWS>
struct StageR {
    static constexpr int RPAD = ROWS + 4;
    static constexpr int CHUNKS = BK * (ROWS / 4);
    static constexpr int PER_THREAD = (CHUNKS + 255) / 256;
    const float* base[PER_THREAD];      // &src[kr][rr], rr = row clamped to nrows-4
    int64_t ldk;
    int shift[PER_THREAD];              // row - rr  (0 except in the chunk that straddles the last row)
    bool any_shift;
    f4u r[PER_THREAD];
    int meta[PER_THREAD];
    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int row0, int nrows, int tid) {
        ldk = ld;
        any_shift = row0 + ROWS > nrows;                 // block-uniform: this tile holds the last rows
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = min(tid + i * 256, CHUNKS - 1);
            const int row = row0 + (c % (ROWS / 4)) * 4;
            const int rr = max(min(row, nrows - 4), 0);
            shift[i] = row < nrows ? row - rr : 4;
            base[i] = src + (int64_t)(c / (ROWS / 4)) * ld + rr;
        }
    }
    __device__ __forceinline__ bool row_edge() const { return any_shift; }
    template <int i>
    __device__ __forceinline__ void load_chunk(int kl, int kseg, bool k_edge, int tid) {
        if (!k_edge) {
            r[i] = *reinterpret_cast<const f4u*>(base[i] + (int64_t)kl * ldk);
            meta[i] = shift[i];
        } else {
            const int c = min(tid + i * 256, CHUNKS - 1);
            const int kr = c / (ROWS / 4);
            const int k = kl + kr;
            const int kc = min(k, kseg - 1);
            meta[i] = k < kseg ? shift[i] : 4;
            r[i] = *reinterpret_cast<const f4u*>(base[i] + (int64_t)(kc - kr) * ldk);
        }
    }
    template <int i>
    __device__ __forceinline__ void store_chunk(float* __restrict__ lds, bool edge, int tid) const {
        const int c = tid + i * 256;
        if (CHUNKS % 256 == 0 || c < CHUNKS) {
            float4 v = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
            if (edge) v = shift_chunk(r[i], meta[i]);
            *reinterpret_cast<float4*>(lds + (c / (ROWS / 4)) * RPAD + (c % (ROWS / 4)) * 4) = v;
        }
    }
};

template <int ROWS>
constexpr int lds_floats(bool kcontig) { return kcontig ? ROWS * KPAD : BK * (ROWS + 4); }

// fragment of one 32-row block for k-group kg (8 k values): f[t] is the operand of MFMA t (k = 8kg+4h+t)
template <bool KCONTIG, int ROWS>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int blk_row0, int kg, int lane, float (&f)[4]) {
    const int i = lane & 31, h = lane >> 5;
    if constexpr (KCONTIG) {
        const float4 v = *reinterpret_cast<const float4*>(lds + (blk_row0 + i) * KPAD + kg * 8 + h * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        const float* p = lds + (kg * 8 + h * 4) * (ROWS + 4) + blk_row0 + i;
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = p[t * (ROWS + 4)];
    }
}

// AK: A stored [M][K] (K-contiguous) else [K][M].  BKC: B stored [N][K] (K-contiguous) else [K][N].
template <bool AK, bool BKC, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256)
gemm_f32_mfma_kernel(const GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;          // 32x32 tiles per wave
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    constexpr int A_FLOATS = lds_floats<BM>(AK), B_FLOATS = lds_floats<BN>(BKC);
    // ONE LDS buffer (27-37 KB): 3-4 workgroups stay resident per CU, which hides the two barriers per
    // K stage better than a double-buffered image at 2 workgroups per CU did (measured: see DESIGN.md)
    __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + B_FLOATS];
    auto sAbuf = [&](int) -> float* { return lds; };
    auto sBbuf = [&](int) -> float* { return lds + A_FLOATS; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // tile mapping: consecutive logical blocks walk N first (they share the A row panel)
    // N may be split into two column segments fed by two B matrices (dW = dZ^T [h | ahn] in ONE launch)
    const int seg_cols = p.Bn2 ? p.Nseg : p.N;
    const int seg_tiles = (seg_cols + BN - 1) / BN;
    const int tiles_n = p.Bn2 ? 2 * seg_tiles : seg_tiles, tiles_m = (p.M + BM - 1) / BM;
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    const unsigned lb = gte_xcd_remap(blockIdx.x, ntile);
    const int tm = lb / tiles_n, tn = lb % tiles_n;
    const int nseg = tn / seg_tiles;                       // 0, or 1 for the second B
    const int m0 = tm * BM, n0 = (tn % seg_tiles) * BN;    // n0: column inside the segment
    const float* Bmat = nseg ? p.Bn2 : p.B;
    const int64_t ldbm = nseg ? p.ldbn2 : p.ldb;
    const int split = blockIdx.y;

    const int tiles_seg0 = (p.K1 + BK - 1) / BK, tiles_seg1 = (p.K2 + BK - 1) / BK;
    const int total_tiles = tiles_seg0 + tiles_seg1;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(total_tiles, t_begin + p.tiles_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    typename std::conditional<AK, StageK<BM>, StageR<BM>>::type stA;
    typename std::conditional<BKC, StageK<BN>, StageR<BN>>::type stB;

    // K segments: 0 = A1 / B rows-or-cols [0,K1), 1 = A2 / B [K1,K1+K2).  The hoisted pointers are
    // (re)initialised when the prefetch crosses into segment 1 (at most once per block).
    constexpr int NM = 4 * TM * TN;                       // MFMAs per k-group
    constexpr int NCA = decltype(stA)::PER_THREAD, NCB = decltype(stB)::PER_THREAD;
    int cur_seg = -1;
    // tile being prefetched: position inside its segment, and whether it is that segment's K-tail tile
    int pf_kl = 0, pf_kseg = 0;
    bool pf_edge = false;       // K-tail flag of the tile whose loads are issued next
    bool st_edge = false;       // ... of the tile sitting in the staging registers (next to be stored)
    auto prefetch_setup = [&](int t) {                    // uniform scalar work + rare pointer re-init
        const int seg = t >= tiles_seg0 ? 1 : 0;
        if (seg != cur_seg) {
            cur_seg = seg;
            const int kb = seg ? p.K1 : 0;
            stA.init(seg ? p.A2 : p.A1, seg ? p.lda2 : p.lda1, m0, p.M, tid);
            if constexpr (BKC) stB.init(Bmat + kb, ldbm, n0, seg_cols, tid);
            else stB.init(Bmat + (int64_t)kb * ldbm, ldbm, n0, seg_cols, tid);
        }
        pf_kl = (seg ? t - tiles_seg0 : t) * BK;
        pf_kseg = seg ? p.K2 : p.K1;
        pf_edge = pf_kl + BK > pf_kseg;
    };
    auto load_chunk = [&](auto J) {                       // chunk J of the tile set up by prefetch_setup
        constexpr int j = decltype(J)::value;
        if constexpr (j < NCA) stA.template load_chunk<j>(pf_kl, pf_kseg, pf_edge, tid);
        else stB.template load_chunk<j - NCA>(pf_kl, pf_kseg, pf_edge, tid);
    };
    auto store_chunk = [&](auto J, int buf) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < NCA) stA.template store_chunk<j>(sAbuf(buf), st_edge || stA.row_edge(), tid);
        else stB.template store_chunk<j - NCA>(sBbuf(buf), st_edge || stB.row_edge(), tid);
    };
    float fa[2][TM][4], fb[2][TN][4];
    auto read_frags = [&](int buf, int kg, int slot) {
#pragma unroll
        for (int a = 0; a < TM; ++a) read_frag<AK, BM>(sAbuf(buf), (wm * TM + a) * 32, kg, lane, fa[slot][a]);
#pragma unroll
        for (int b = 0; b < TN; ++b) read_frag<BKC, BN>(sBbuf(buf), (wn * TN + b) * 32, kg, lane, fb[slot][b]);
    };
    auto mfma_one = [&](int slot, int j) {                // j -> (tt, a, b), b fastest: consecutive MFMAs hit different accumulators
        const int b = j % TN, a = (j / TN) % TM, tt = j / (TN * TM);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][tt], fb[slot][b][tt], acc[a][b], 0, 0, 0);
    };
    auto mfma_group = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NM; ++j) mfma_one(slot, j);
    };

    // Per K stage: barrier | staging registers (tile t) -> LDS | barrier | issue the global loads of tile
    // t+1 (they land under this stage's MFMAs) | 4 k-groups of MFMAs, the fragments of k-group g+1 being
    // read while the MFMAs of k-group g run (sched_barrier pins "reads before the MFMAs they hide under").
    // Tried and measured slower or equal on the page-batch shapes (profiles/r01/gemm_variants.md): LDS double
    // buffering with one barrier per stage (2 workgroups/CU instead of 3-4), per-MFMA interleaving of the
    // staging instructions, a start-time stagger of co-resident workgroups.
    if (t_begin < t_end) {
        prefetch_setup(t_begin);
        static_for<NCA + NCB>([&](auto J) { load_chunk(J); });
    }
    for (int t = t_begin; t < t_end; ++t) {
        st_edge = pf_edge;
        __syncthreads();                       // previous tile's fragment reads are done
        static_for<NCA + NCB>([&](auto J) { store_chunk(J, 0); });
        __syncthreads();
        if (t + 1 < t_end) {                   // prefetch under the MFMAs
            prefetch_setup(t + 1);
            static_for<NCA + NCB>([&](auto J) { load_chunk(J); });
        }
        read_frags(0, 0, 0);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            if (kg + 1 < BK / 8) read_frags(0, kg + 1, (kg + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(kg & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    float* outp = p.splits > 1 ? p.slab + (int64_t)split * p.M * p.N : p.C;
    const int64_t ldo = p.splits > 1 ? p.N : p.ldc;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col_in_seg = n0 + (wn * TN + b) * 32 + col_l;
        if (col_in_seg >= seg_cols) continue;
        const int col = nseg * seg_cols + col_in_seg;
        const float bv = (p.bias && p.splits <= 1) ? p.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int rbase = m0 + (wm * TM + a) * 32 + hrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    float v = acc[a][b][r] + bv;
                    float* dst = outp + (int64_t)row * ldo + col;
                    if (p.splits <= 1) {
                        if (p.accumulate) v += *dst;
                        if (p.relu) v = fmaxf(v, 0.f);
                    }
                    *dst = v;
                }
            }
        }
    }
}

// C[m][n] (+)= bias[n] + sum_s slab[s][m][n]   -- fixed order, deterministic.
// A block folds 32 consecutive output elements; its 8 split-lanes each sum every 8th slab (independent
// loads in flight, 128-B coalesced over the elements), then the 8 partials are added in lane order through
// LDS.  (First version: one thread per element looping over all slabs -- with a 9 x 256 output and hundreds
// of slabs that was 9 workgroups chasing one load at a time: 18 us per launch, 9 launches per step.)
constexpr int RED_E = 32, RED_S = 8;
__global__ void __launch_bounds__(RED_E * RED_S)
splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t mn, int N, float* __restrict__ C, int64_t ldc,
                     const float* __restrict__ bias, int accumulate, int relu) {
    __shared__ float part[RED_S][RED_E];
    const int e = threadIdx.x % RED_E, sl = threadIdx.x / RED_E;
    const int64_t i = (int64_t)blockIdx.x * RED_E + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < mn) {
        int k = sl;
#pragma unroll 4
        for (; k + RED_S < splits; k += 2 * RED_S) {             // 8 independent loads in flight
            s0 += slab[(int64_t)k * mn + i];
            s1 += slab[(int64_t)(k + RED_S) * mn + i];
        }
        if (k < splits) s0 += slab[(int64_t)k * mn + i];
    }
    part[sl][e] = s0 + s1;
    __syncthreads();
    if (sl == 0 && i < mn) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < RED_S; ++q) s += part[q][e];
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        if (bias) s += bias[n];
        float* dst = C + m * ldc + n;
        if (accumulate) s += *dst;
        if (relu) s = fmaxf(s, 0.f);
        *dst = s;
    }
}

// ------------------------------- skinny GEMMs (class-count-sized dimension) -----------------------
// The last layer has n_classes = 9 outputs.  Its dW = dZ^T X is a [9 x F] result reduced over all nodes and
// its dX = dZ W has an inner dimension of 9: both are HBM-bound (read or write one [N, F] matrix) with
// ~18 flop/byte, and a 32-wide MFMA tile would spend 72 % of its rows on padding.  Plain fp32 FMA chains.
constexpr int SK_MAX = 16;         // skinny dimension bound
constexpr int SK_ROWS = 32;        // k rows (nodes) per block of the TN kernel: ~N/32 blocks fill the chip

// C[m][n] = sum_k A[k][m] * B[k][n],  m < M <= 16;  A stored [K][M] (lda), B stored [K][N] (ldb).
// One thread per column n (coalesced B rows, all SK_ROWS loads of a thread in flight at once); the block's
// A rows are staged in LDS and read back as broadcasts.  Partial sums of the block's k-chunk go to
// slab[blockIdx.y][m][n]; splitk_reduce_kernel folds them in a fixed order (deterministic, no atomics).
__global__ void __launch_bounds__(256)
gemm_tn_skinny_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                      float* __restrict__ slab, int M, int N, int K) {
    __shared__ float sA[SK_ROWS][SK_MAX];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * SK_ROWS, nk = min(SK_ROWS, K - k0);
    for (int i = threadIdx.x; i < SK_ROWS * SK_MAX; i += 256) {
        const int r = i / SK_MAX, c = i % SK_MAX;
        sA[r][c] = (r < nk && c < M) ? A[(int64_t)(k0 + r) * lda + c] : 0.f;
    }
    float b[SK_ROWS];
    const int nn = min(n, N - 1);
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) b[r] = B[(int64_t)(k0 + min(r, nk - 1)) * ldb + nn];
    __syncthreads();
    float acc[SK_MAX];
#pragma unroll
    for (int i = 0; i < SK_MAX; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) {
#pragma unroll
        for (int i = 0; i < SK_MAX; ++i) acc[i] = fmaf(sA[r][i], b[r], acc[i]);   // rows >= nk hold zeros in sA
    }
    if (n < N) {
        float* out = slab + (int64_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < SK_MAX; ++i)
            if (i < M) out[(int64_t)i * N + n] = acc[i];
    }
}

// C[m][n] (+)= sum_k A[m][k] * B[k][n],  K <= 16;  A stored [M][K] (lda), B stored [K][N] (ldb).
// One thread per 4 consecutive columns (16-byte loads of B and stores of C, 4-byte aligned is enough on
// gfx950); the K x N operand B is small (<= 16 rows) and stays in L1/L2; A's row is a wave-wide broadcast.
__global__ void __launch_bounds__(256)
gemm_nn_skinny_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                      float* __restrict__ C, int64_t ldc, int M, int N, int K, int accumulate) {
    const int nq = (N + 3) / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * nq) return;
    const int m = (int)(idx / nq), n = (int)(idx % nq) * 4;
    const float* a = A + (int64_t)m * lda;
    float* c = C + (int64_t)m * ldc + n;
    if (n + 3 < N) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
            const float av = a[k];
            const f4u b = *reinterpret_cast<const f4u*>(B + (int64_t)k * ldb + n);
            acc.x = fmaf(av, b.x, acc.x); acc.y = fmaf(av, b.y, acc.y);
            acc.z = fmaf(av, b.z, acc.z); acc.w = fmaf(av, b.w, acc.w);
        }
        if (accumulate) { const f4u o = *reinterpret_cast<const f4u*>(c); acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        f4u o; o.x = acc.x; o.y = acc.y; o.z = acc.z; o.w = acc.w;
        *reinterpret_cast<f4u*>(c) = o;
    } else {
        for (int j = 0; n + j < N; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(a[k], B[(int64_t)k * ldb + n + j], acc);
            c[j] = accumulate ? c[j] + acc : acc;
        }
    }
}

// Shapes with a contiguous run shorter than 4 floats (F = 3 toy graphs, a single-node page): one thread
// per output element, plain fp32 FMA chain in k order.  Never on the measured path.
template <bool AK, bool BKC>
__global__ void __launch_bounds__(256)
gemm_small_kernel(const GemmParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.M * p.N) return;
    const int m = (int)(idx / p.N), n = (int)(idx % p.N);
    float acc = 0.f;
    for (int seg = 0; seg < 2; ++seg) {
        const float* A = seg ? p.A2 : p.A1;
        const int64_t lda = seg ? p.lda2 : p.lda1;
        const int K = seg ? p.K2 : p.K1, kb = seg ? p.K1 : 0;
        for (int k = 0; k < K; ++k) {
            const float a = AK ? A[(int64_t)m * lda + k] : A[(int64_t)k * lda + m];
            const float* Bm = (p.Bn2 && n >= p.Nseg) ? p.Bn2 : p.B;
            const int64_t ldbm = (p.Bn2 && n >= p.Nseg) ? p.ldbn2 : p.ldb;
            const int nn = (p.Bn2 && n >= p.Nseg) ? n - p.Nseg : n;
            const float b = BKC ? Bm[(int64_t)nn * ldbm + kb + k] : Bm[(int64_t)(kb + k) * ldbm + nn];
            acc = fmaf(a, b, acc);
        }
    }
    if (p.bias) acc += p.bias[n];
    float* dst = p.C + (int64_t)m * p.ldc + n;
    if (p.accumulate) acc += *dst;
    if (p.relu) acc = fmaxf(acc, 0.f);
    *dst = acc;
}

// every contiguous run the MFMA kernel's 16-byte staging loads touch must hold >= 4 floats
bool needs_small_path(bool ak, bool bkc, const GemmParams& p) {
    const int kmin = p.K2 > 0 ? (p.K1 < p.K2 ? p.K1 : p.K2) : p.K1;
    const int a_run = ak ? kmin : p.M;
    const int b_run = bkc ? kmin : (p.Bn2 ? p.Nseg : p.N);
    return a_run < 4 || b_run < 4;
}

struct Plan { int bm, bn, tiles, splits, tiles_per_split; };

Plan make_plan(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t Nseg = 0) {
    Plan pl;
    const int cus = gte::device_props().cus;
    if (N <= 32) { pl.bm = 128; pl.bn = 32; }
    else if (M <= 32) { pl.bm = 32; pl.bn = 128; }
    else {
        // tile quantisation: a page batch gives ~1.5 waves of 128x128 tiles over the 256 CUs (the last
        // half-empty wave costs a full tile time); 64x128 tiles cut the tail.  Compare rounds x tile area.
        const int64_t t128 = gte::ceil_div(M, 128) * gte::ceil_div(N, 128);
        const int64_t t64 = gte::ceil_div(M, 64) * gte::ceil_div(N, 128);
        const int64_t c128 = gte::ceil_div(t128, cus) * 2, c64 = gte::ceil_div(t64, cus);
        pl.bn = 128;
        pl.bm = (t128 >= cus && c64 * 10 < c128 * 9) ? 64 : 128;
        // split-K with very few output tiles (dW of a 256 x 256 layer: 4 tiles over 24 k nodes): smaller tiles ->
        // half the K splits -> half the slab bytes (measured 58 -> 44 us; at 14 tiles, 256 x 831, it loses: 147 -> 161)
        if (t128 <= 8 && M >= 64) pl.bm = 64;
    }
    pl.tiles = (int)(gte::ceil_div(M, pl.bm) * (Nseg > 0 ? 2 * gte::ceil_div(Nseg, pl.bn) : gte::ceil_div(N, pl.bn)));
    const int ktiles = (int)(gte::ceil_div(K1, BK) + gte::ceil_div(K2, BK));
    int splits = 1;
    // the reduction dimension is the node count for dW = dZ^T X: few output tiles, very long K
    if (pl.tiles < cus && ktiles >= 16) {
        // fill the resident slots (2 workgroups per CU) EXACTLY or stay below: 28 tiles x 19 splits = 532 workgroups
        // on 512 slots ran a 20-workgroup straggler round that cost a third of the kernel (305 us -> see DESIGN)
        splits = (2 * cus) / pl.tiles;
        const int max_splits = ktiles / 8 > 0 ? ktiles / 8 : 1;      // >= 8 K tiles (256 k) per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    pl.tiles_per_split = (int)gte::ceil_div(ktiles, splits);
    pl.splits = (int)gte::ceil_div(ktiles, pl.tiles_per_split);
    return pl;
}

template <bool AK, bool BKC>
int launch_shape(const GemmParams& p, const Plan& pl, hipStream_t s) {
    dim3 grid((unsigned)pl.tiles, (unsigned)pl.splits), block(256);
    if (pl.bm == 128 && pl.bn == 128)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 128, 128, 2, 2>), grid, block, 0, s, p);
    else if (pl.bm == 64 && pl.bn == 128)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 64, 128, 2, 2>), grid, block, 0, s, p);
    else if (pl.bn == 32)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 128, 32, 4, 1>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 32, 128, 1, 4>), grid, block, 0, s, p);
    return gte::check_launch("gemm_f32");
}

int run_small(bool ak, bool bkc, GemmParams p, hipStream_t s) {
    p.splits = 1;
    dim3 grid((unsigned)gte::ceil_div((int64_t)p.M * p.N, 256)), block(256);
    if (ak && bkc) hipLaunchKernelGGL((gemm_small_kernel<true, true>), grid, block, 0, s, p);
    else if (ak) hipLaunchKernelGGL((gemm_small_kernel<true, false>), grid, block, 0, s, p);
    else if (bkc) hipLaunchKernelGGL((gemm_small_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_small_kernel<false, false>), grid, block, 0, s, p);
    return gte::check_launch("gemm_small");
}

int run_gemm(bool ak, bool bkc, GemmParams p, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    if (p.M == 0 || p.N == 0) return GTE_OK;
    if (needs_small_path(ak, bkc, p)) return run_small(ak, bkc, p, s);
    const Plan pl = make_plan(p.M, p.N, p.K1, p.K2, p.Bn2 ? p.Nseg : 0);
    p.splits = pl.splits;
    p.tiles_per_split = pl.tiles_per_split;
    p.slab = nullptr;
    const int relu = p.relu, accumulate = p.accumulate;
    if (pl.splits > 1) {
        const int64_t need = (int64_t)pl.splits * p.M * p.N * (int64_t)sizeof(float);
        if (!workspace || workspace_bytes < need)
            return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gemm_f32: split-K needs %lld workspace bytes, got %lld",
                             (long long)need, (long long)workspace_bytes);
        p.slab = reinterpret_cast<float*>(workspace);
    }
    int rc;
    if (ak && bkc) rc = launch_shape<true, true>(p, pl, s);
    else if (ak && !bkc) rc = launch_shape<true, false>(p, pl, s);
    else if (!ak && bkc) rc = launch_shape<false, true>(p, pl, s);
    else rc = launch_shape<false, false>(p, pl, s);
    if (rc != GTE_OK) return rc;
    if (pl.splits > 1) {
        const int64_t mn = (int64_t)p.M * p.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gte::ceil_div(mn, RED_E)), dim3(RED_E * RED_S), 0, s, p.slab,
                           pl.splits, mn, p.N, p.C, p.ldc, p.bias, accumulate, relu);
        return gte::check_launch("gemm_f32 split-K reduce");
    }
    return GTE_OK;
}

int64_t gemm_workspace(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t Nseg = 0) {
    if (M <= 0 || N <= 0) return 256;
    const Plan pl = make_plan(M, N, K1, K2, Nseg);
    return pl.splits > 1 ? gte::round_up((int64_t)pl.splits * M * N * 4, 256) : 256;
}

// ------------------------------- LayerNorm + ReLU, forward ----------------------------------------
// y = relu?( gamma * (z - mean) * rstd + beta ); one wave per row, two-pass mean/variance (biased
// variance, eps inside the sqrt: torch.nn.LayerNorm).  In place (y == z) is allowed.
constexpr int LN_CACHE = 16;      // elements per lane kept in registers -> rows up to 1024 wide

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float ln_affine(float xhat, float g, float b) { return fmaf(xhat, g, b); }

__global__ void __launch_bounds__(256)
ln_relu_fwd_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, int relu, float* __restrict__ y, int64_t ldy,
                   float* __restrict__ stats, int M, int n) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* zr = z + (int64_t)row * ldz;
    float* yr = y + (int64_t)row * ldy;
    float c[LN_CACHE];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_CACHE; ++t) {
        const int j = lane + 64 * t;
        c[t] = j < n ? zr[j] : 0.f;
        s += c[t];
    }
    for (int j = lane + 64 * LN_CACHE; j < n; j += 64) s += zr[j];
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_CACHE; ++t) {
        const int j = lane + 64 * t;
        const float d = j < n ? c[t] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    for (int j = lane + 64 * LN_CACHE; j < n; j += 64) { const float d = zr[j] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)n + eps);
    if (stats && lane == 0) { stats[row] = mean; stats[M + row] = rstd; }
#pragma unroll
    for (int t = 0; t < LN_CACHE; ++t) {
        const int j = lane + 64 * t;
        if (j < n) {
            float v = ln_affine((c[t] - mean) * rstd, gamma[j], beta[j]);
            if (relu) v = fmaxf(v, 0.f);
            yr[j] = v;
        }
    }
    for (int j = lane + 64 * LN_CACHE; j < n; j += 64) {
        float v = ln_affine((zr[j] - mean) * rstd, gamma[j], beta[j]);
        if (relu) v = fmaxf(v, 0.f);
        yr[j] = v;
    }
}

// ------------------------------- LayerNorm + ReLU, backward ---------------------------------------
// Per row: g = relu ? (pre > 0 ? dy : 0) : dy ; dxhat = g*gamma ;
//          dz = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat))
// Column sums (dgamma = sum g*xhat, dbeta = sum g, dbias = sum dz) never touch atomics: every wave
// keeps its partial sums for its columns (lane + 64 t) in registers over the rows it owns, the four
// waves of a block are folded through LDS, the block writes partial[block][3][n], and a second kernel
// folds the (<= LNB_MAX_BLOCKS) block partials in a fixed order.  HBM-bound: reads dy and z once,
// writes dz once.
constexpr int LNB_MAX_BLOCKS = 512;

template <int NCH>                 // row width n <= 64 * NCH, whole row in registers
__global__ void __launch_bounds__(256)
ln_relu_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z, int64_t ldz,
                   const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                   int relu, float* __restrict__ dz, int64_t lddz, float* __restrict__ partial, int M, int n) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][3][NCH*64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool has_ln = gamma != nullptr;
    float gam[NCH], bet[NCH], s_dg[NCH], s_db[NCH], s_dbias[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        const int j = lane + 64 * t;
        gam[t] = (has_ln && j < n) ? gamma[j] : 1.f;
        bet[t] = (has_ln && j < n) ? beta[j] : 0.f;
        s_dg[t] = s_db[t] = s_dbias[t] = 0.f;
    }
    const float inv_n = 1.0f / (float)n;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float* dyr = dy + (int64_t)row * lddy;
        const float* zr = z + (int64_t)row * ldz;
        float* dzr = dz + (int64_t)row * lddz;
        float g[NCH], xh[NCH];
        if (has_ln) {
            const float mean = stats[row], rstd = stats[M + row];
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const int j = lane + 64 * t;
                const bool ok = j < n;
                xh[t] = ok ? (zr[j] - mean) * rstd : 0.f;
                float gv = ok ? dyr[j] : 0.f;
                if (relu && ln_affine(xh[t], gam[t], bet[t]) <= 0.f) gv = 0.f;
                g[t] = gv;
                const float dxh = gv * gam[t];
                a += dxh;
                b = fmaf(dxh, xh[t], b);
            }
            const float c1 = wave_sum(a) * inv_n, c2 = wave_sum(b) * inv_n;
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const int j = lane + 64 * t;
                if (j < n) {
                    const float d = rstd * (g[t] * gam[t] - c1 - xh[t] * c2);
                    s_dg[t] = fmaf(g[t], xh[t], s_dg[t]);
                    s_db[t] += g[t];
                    s_dbias[t] += d;
                    dzr[j] = d;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const int j = lane + 64 * t;
                if (j < n) {
                    float gv = dyr[j];
                    if (relu && zr[j] <= 0.f) gv = 0.f;
                    s_dbias[t] += gv;
                    dzr[j] = gv;
                }
            }
        }
    }
    constexpr int W = NCH * 64;
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        red[(wave * 3 + 0) * W + lane + 64 * t] = s_dg[t];
        red[(wave * 3 + 1) * W + lane + 64 * t] = s_db[t];
        red[(wave * 3 + 2) * W + lane + 64 * t] = s_dbias[t];
    }
    __syncthreads();
    float* pp = partial + (int64_t)blockIdx.x * 3 * n;
    for (int i = threadIdx.x; i < 3 * W; i += 256) {
        const int q = i / W, j = i - q * W;
        if (j < n) pp[q * n + j] = red[(0 * 3 + q) * W + j] + red[(1 * 3 + q) * W + j] + red[(2 * 3 + q) * W + j] +
                                   red[(3 * 3 + q) * W + j];
    }
}

// Rows wider than 1024: same maths, the row is re-read from L1/L2 instead of cached in registers, and
// the column partials are produced 64 columns at a time.  dz must not alias dy here.
__global__ void __launch_bounds__(256)
ln_relu_bwd_wide_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z, int64_t ldz,
                        const float* __restrict__ stats, const float* __restrict__ gamma,
                        const float* __restrict__ beta, int relu, float* __restrict__ dz, int64_t lddz,
                        float* __restrict__ partial, int M, int n) {
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool has_ln = gamma != nullptr;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool jok = j < n;
        const float gj = (has_ln && jok) ? gamma[j] : 1.f;
        const float bj = (has_ln && jok) ? beta[j] : 0.f;
        float s_dg = 0.f, s_db = 0.f, s_dbias = 0.f;
        for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
            const float* dyr = dy + (int64_t)row * lddy;
            const float* zr = z + (int64_t)row * ldz;
            float* dzr = dz + (int64_t)row * lddz;
            if (has_ln) {
                const float mean = stats[row], rstd = stats[M + row];
                float a = 0.f, b = 0.f;
                for (int jj = lane; jj < n; jj += 64) {
                    const float xh = (zr[jj] - mean) * rstd;
                    float g = dyr[jj];
                    if (relu && ln_affine(xh, gamma[jj], beta[jj]) <= 0.f) g = 0.f;
                    const float dxh = g * gamma[jj];
                    a += dxh;
                    b = fmaf(dxh, xh, b);
                }
                const float c1 = wave_sum(a) / (float)n, c2 = wave_sum(b) / (float)n;
                if (jok) {
                    const float xh = (zr[j] - mean) * rstd;
                    float g = dyr[j];
                    if (relu && ln_affine(xh, gj, bj) <= 0.f) g = 0.f;
                    const float d = rstd * (g * gj - c1 - xh * c2);
                    s_dg = fmaf(g, xh, s_dg);
                    s_db += g;
                    s_dbias += d;
                    dzr[j] = d;
                }
            } else if (jok) {
                float g = dyr[j];
                if (relu && zr[j] <= 0.f) g = 0.f;
                s_dbias += g;
                dzr[j] = g;
            }
        }
        red[0][wave][lane] = s_dg; red[1][wave][lane] = s_db; red[2][wave][lane] = s_dbias;
        __syncthreads();
        if (wave == 0 && jok) {
            float* pp = partial + (int64_t)blockIdx.x * 3 * n;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                pp[q * n + j] = red[q][0][lane] + red[q][1][lane] + red[q][2][lane] + red[q][3][lane];
        }
        __syncthreads();
    }
}

// fold the block partials: block = 64 columns x 16 slices of the block list; fixed order; WRITES results
__global__ void __launch_bounds__(1024)
colsum_fold_kernel(const float* __restrict__ partial, int nblocks, int n, float* __restrict__ dgamma,
                   float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f, c = 0.f;
    if (j < n) {
#pragma unroll 4
        for (int k = slice; k < nblocks; k += 16) {               // 4 x 3 independent loads in flight
            const float* pp = partial + (int64_t)k * 3 * n;
            a += pp[j]; b += pp[n + j]; c += pp[2 * n + j];
        }
    }
    red[0][slice][lane] = a; red[1][slice][lane] = b; red[2][slice][lane] = c;
    __syncthreads();
    if (slice < 3 && j < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[slice][k][lane];
        float* dst = slice == 0 ? dgamma : slice == 1 ? dbeta : dbias;
        if (dst) dst[j] = s;
    }
}

int ln_bwd_blocks(int64_t M) {
    const int64_t b = gte::ceil_div(M, 4);
    return (int)(b < LNB_MAX_BLOCKS ? b : LNB_MAX_BLOCKS);
}

}  // namespace

// ------------------------------------------ C ABI -------------------------------------------------
extern "C" int64_t gte_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t mfma = gemm_workspace(M, N, K, 0);
    const int64_t skinny = (M > 0 && M <= SK_MAX) ? gte::round_up(gte::ceil_div(K > 0 ? K : 1, SK_ROWS) * M * N * 4, 256) : 0;
    return mfma > skinny ? mfma : skinny;
}

extern "C" int gte_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_f32: bad sizes");
    if (M == 0 || N == 0) return GTE_OK;
    if (!A || !B || !C) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_f32: null pointer");
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_f32: leading dimension too small");
    hipStream_t s = gte::as_stream(stream);
    if (trans_a && !trans_b && M <= SK_MAX && K >= 4 * SK_ROWS) {          // dW of a class-count-sized layer
        const int splits = (int)gte::ceil_div(K, SK_ROWS);
        const int64_t need = (int64_t)splits * M * N * 4;
        if (!workspace || workspace_bytes < need)
            return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gemm_f32 (skinny TN): workspace %lld < %lld",
                             (long long)workspace_bytes, (long long)need);
        float* slab = reinterpret_cast<float*>(workspace);
        hipLaunchKernelGGL(gemm_tn_skinny_kernel, dim3((unsigned)gte::ceil_div(N, 256), (unsigned)splits), dim3(256), 0, s,
                           A, lda, B, ldb, slab, (int)M, (int)N, (int)K);
        const int64_t mn = M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gte::ceil_div(mn, RED_E)), dim3(RED_E * RED_S), 0, s, slab,
                           splits, mn, (int)N, C, ldc, (const float*)nullptr, accumulate ? 1 : 0, 0);
        return gte::check_launch("gemm_f32 skinny TN");
    }
    if (!trans_a && !trans_b && K <= SK_MAX) {                              // dX through a class-count-sized layer
        const int64_t work = M * gte::ceil_div(N, 4);
        hipLaunchKernelGGL(gemm_nn_skinny_kernel, dim3((unsigned)gte::ceil_div(work, 256)), dim3(256), 0, s, A, lda, B, ldb, C,
                           ldc, (int)M, (int)N, (int)K, accumulate ? 1 : 0);
        return gte::check_launch("gemm_f32 skinny NN");
    }
    GemmParams p = {};
    p.A1 = A; p.lda1 = lda; p.K1 = (int)K; p.A2 = nullptr; p.lda2 = 0; p.K2 = 0;
    p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.bias = nullptr; p.M = (int)M; p.N = (int)N;
    p.relu = 0; p.accumulate = accumulate ? 1 : 0;
    return run_gemm(!trans_a, trans_b != 0, p, workspace, workspace_bytes, s);
}

extern "C" int64_t gte_sage_linear_dw_workspace_bytes(int64_t n_out, int64_t k1, int64_t k2, int64_t n_nodes) {
    if (k2 > 0 && k2 == k1 && n_out > SK_MAX) return gemm_workspace(n_out, 2 * k1, n_nodes, 0, k1);
    const int64_t a = gte_gemm_workspace_bytes(n_out, k1, n_nodes), b = k2 > 0 ? gte_gemm_workspace_bytes(n_out, k2, n_nodes) : 0;
    return a > b ? a : b;
}

extern "C" int gte_sage_linear_dw(const float* dz, int64_t lddz, const float* x1, int64_t ldx1, int64_t k1,
                                  const float* x2, int64_t ldx2, int64_t k2, float* dW, int64_t lddw, int64_t n_out,
                                  int64_t n_nodes, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_out <= 0 || k1 <= 0 || k2 < 0 || n_nodes < 0 || n_out > INT32_MAX || k1 + k2 > INT32_MAX || n_nodes > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_dw: bad sizes");
    if (!dz || !x1 || !dW || (k2 > 0 && !x2)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_dw: null pointer");
    if (lddz < n_out || ldx1 < k1 || (k2 > 0 && ldx2 < k2) || lddw < k1 + k2)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_dw: leading dimension too small");
    if (n_nodes == 0) {
        for (int64_t r = 0; r < n_out; ++r)
            if (hipMemsetAsync(dW + r * lddw, 0, (size_t)(k1 + k2) * 4, gte::as_stream(stream)) != hipSuccess)
                return gte::fail(GTE_ERR_LAUNCH, "sage_linear_dw: memset failed");
        return GTE_OK;
    }
    const bool skinny = n_out <= SK_MAX;
    if (k2 > 0 && k2 == k1 && !skinny) {
        // one launch: C[n_out, 2F] = dZ^T [x1 | x2]; A = dZ stored [nodes][n_out], B segments stored [nodes][F]
        GemmParams p = {};
        p.A1 = dz; p.lda1 = lddz; p.K1 = (int)n_nodes; p.A2 = nullptr; p.lda2 = 0; p.K2 = 0;
        p.B = x1; p.ldb = ldx1; p.Bn2 = x2; p.ldbn2 = ldx2; p.Nseg = (int)k1;
        p.C = dW; p.ldc = lddw; p.bias = nullptr; p.M = (int)n_out; p.N = (int)(2 * k1); p.relu = 0; p.accumulate = 0;
        return run_gemm(false, false, p, workspace, workspace_bytes, gte::as_stream(stream));
    }
    int rc = gte_gemm_f32(1, 0, n_out, k1, n_nodes, dz, lddz, x1, ldx1, dW, lddw, 0, workspace, workspace_bytes, stream);
    if (rc != GTE_OK || k2 == 0) return rc;
    return gte_gemm_f32(1, 0, n_out, k2, n_nodes, dz, lddz, x2, ldx2, dW + k1, lddw, 0, workspace, workspace_bytes, stream);
}

extern "C" int gte_sage_linear_fwd(const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2,
                                   int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                   const float* beta, float eps, int relu, float* z_save, int64_t ldz, float* stats,
                                   float* y, int64_t ldy, int64_t M, int64_t n_out, void* stream) {
    if (M < 0 || n_out <= 0 || k1 <= 0 || k2 < 0 || M > INT32_MAX || n_out > INT32_MAX || k1 + k2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: bad sizes");
    if (M == 0) return GTE_OK;
    if (!a1 || !W || !y || (k2 > 0 && !a2)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: null pointer");
    if (lda1 < k1 || (k2 > 0 && lda2 < k2) || ldw < k1 + k2 || ldy < n_out || (z_save && ldz < n_out))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: leading dimension too small");
    if (gamma && !beta) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: gamma without beta");
    hipStream_t s = gte::as_stream(stream);
    const bool ln = gamma != nullptr;
    // z goes to z_save when the backward needs it, else straight into y (LayerNorm then runs in place)
    float* zbuf = (ln && z_save) ? z_save : y;
    const int64_t ldzz = (ln && z_save) ? ldz : ldy;
    GemmParams p = {};
    p.A1 = a1; p.lda1 = lda1; p.K1 = (int)k1; p.A2 = a2; p.lda2 = lda2; p.K2 = (int)k2;
    p.B = W; p.ldb = ldw; p.C = zbuf; p.ldc = ldzz; p.bias = bias; p.M = (int)M; p.N = (int)n_out;
    p.relu = (!ln && relu) ? 1 : 0; p.accumulate = 0;
    // forward K = k1+k2 is short and M is the node count: never split (workspace-free)
    const Plan pl = make_plan(M, n_out, k1, k2);
    if (needs_small_path(true, true, p)) {
        int rc = run_small(true, true, p, s);
        if (rc != GTE_OK) return rc;
    } else if (pl.splits > 1) {
        // tiny M with long K (not a page-graph shape): run unsplit rather than demand a workspace
        Plan one = pl; one.splits = 1; one.tiles_per_split = (int)(gte::ceil_div(k1, BK) + gte::ceil_div(k2, BK));
        p.splits = 1; p.tiles_per_split = one.tiles_per_split;
        int rc = launch_shape<true, true>(p, one, s);
        if (rc != GTE_OK) return rc;
    } else {
        int rc = run_gemm(true, true, p, nullptr, 0, s);
        if (rc != GTE_OK) return rc;
    }
    if (!ln && z_save && z_save != y) {
        // no LayerNorm but the caller wants z (= pre-activation): only meaningful with relu; keep the contract simple
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_linear_fwd: z_save without LayerNorm is not supported");
    }
    if (ln) {
        hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3((unsigned)gte::ceil_div(M, 4)), dim3(256), 0, s, zbuf, ldzz, gamma,
                           beta, eps, relu, y, ldy, stats, (int)M, (int)n_out);
        return gte::check_launch("ln_relu_fwd");
    }
    return GTE_OK;
}

extern "C" int gte_ln_relu_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu,
                               float* y, int64_t ldy, float* stats, int64_t M, int64_t n_out, void* stream) {
    if (M < 0 || n_out <= 0 || M > INT32_MAX || n_out > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: bad sizes");
    if (M == 0) return GTE_OK;
    if (!z || !y || !gamma || !beta) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: null pointer");
    if (ldz < n_out || ldy < n_out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: ld < n_out");
    hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3((unsigned)gte::ceil_div(M, 4)), dim3(256), 0, gte::as_stream(stream), z,
                       ldz, gamma, beta, eps, relu, y, ldy, stats, (int)M, (int)n_out);
    return gte::check_launch("ln_relu_fwd");
}

extern "C" int64_t gte_ln_relu_bwd_workspace_bytes(int64_t M, int64_t n_out) {
    return gte::round_up((int64_t)ln_bwd_blocks(M > 0 ? M : 1) * 3 * (n_out > 0 ? n_out : 1) * 4, 256);
}

extern "C" int gte_ln_relu_bwd(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats,
                               const float* gamma, const float* beta, int relu, float* dz, int64_t lddz,
                               float* dgamma, float* dbeta, float* dbias, int64_t M, int64_t n_out, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    if (M < 0 || n_out <= 0 || M > INT32_MAX || n_out > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: bad sizes");
    if (M == 0) return GTE_OK;
    if (!dy || !dz || !workspace) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: null pointer");
    if ((gamma || relu) && !z) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: z is required");
    if (gamma && (!beta || !stats)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: LayerNorm needs beta and stats");
    if (gamma && n_out > 1024 && dz == dy)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: dz must not alias dy when n_out > 1024 with LayerNorm");
    if (workspace_bytes < gte_ln_relu_bwd_workspace_bytes(M, n_out))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "ln_relu_bwd: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    const int nb = ln_bwd_blocks(M);
    const float* zz = z ? z : dy;
    const int64_t ldzz = z ? ldz : lddy;
    float* part = reinterpret_cast<float*>(workspace);
    dim3 grid((unsigned)nb), block(256);
#define GTE_LNB(NCH)                                                                                              \
    hipLaunchKernelGGL((ln_relu_bwd_kernel<NCH>), grid, block, (size_t)(4 * 3 * NCH * 64) * sizeof(float), s, dy, lddy, \
                       zz, ldzz, stats, gamma, beta, relu, dz, lddz, part, (int)M, (int)n_out)
    if (n_out <= 64) GTE_LNB(1);
    else if (n_out <= 128) GTE_LNB(2);
    else if (n_out <= 256) GTE_LNB(4);
    else if (n_out <= 512) GTE_LNB(8);
    else if (n_out <= 1024) GTE_LNB(16);
    else
        hipLaunchKernelGGL(ln_relu_bwd_wide_kernel, grid, block, 0, s, dy, lddy, zz, ldzz, stats, gamma, beta, relu, dz,
                           lddz, part, (int)M, (int)n_out);
#undef GTE_LNB
    if (dgamma || dbeta || dbias)
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)gte::ceil_div(n_out, 64)), dim3(1024), 0, s, part, nb,
                           (int)n_out, dgamma, dbeta, dbias);
    return gte::check_launch("ln_relu_bwd");
}
