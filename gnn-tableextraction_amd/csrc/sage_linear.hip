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
//   block tile BM x BN (128x128, 64x128, 128x32 or 32x128), BK = 32; every wave owns a (BM/WM)x(BN/WN) patch
//   of 32x32 MFMA tiles held in accumulator registers.
//   global -> registers (buffer_load_dwordx4 through a descriptor that range-checks the tile window; tile t+2
//          is in flight while tile t is multiplied) -> LDS double buffer (ds_write_b128) -> fragments.
//   K-contiguous operands ([rows][K] in memory) keep [rows][BK+4] LDS images: the +4 pad makes the
//   16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots; one b128 read feeds 4 MFMAs
//   (k = k0+4h+t, h = lane>>5: any k order is valid as long as A and B agree).
//   Row-contiguous operands ([K][rows]) keep [BK][rows+4] images read with conflict-free ds_read_b32.
//   blockIdx is remapped so that the tiles sharing an A row-panel run on one XCD (its L2 keeps it).
#include "gte_common.h"
#include "p3.h"

#include <stdlib.h>
#include <string.h>
#include <atomic>
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
    const float* B2;                   // optional origin of B's K segment 1 (default: K1 rows / columns after B's origin)
    const float* Bn2; int64_t ldbn2;   // optional second B for output columns [Nseg, 2*Nseg): C = A [B | Bn2]
    int Nseg;                          // columns per N segment when Bn2 != nullptr (then N == 2*Nseg)
    const float* An2; int64_t ldan2;   // optional A of N segment 1 (with Bn2): C = [A B | An2 Bn2]
    int bias_cols;                     // > 0: the bias covers columns [0, bias_cols) only
    // tail split (whole-K launches only): logical tiles [sk_full, ntile) are cut into sk_pieces K ranges of sk_tps K
    // tiles; piece p of tail tile j writes its tile-local partial to sk_slab[(j * sk_pieces + p)][BM][BN]
    int sk_full, sk_pieces, sk_tps;
    float* sk_slab;
    float* C; int64_t ldc;
    const float* bias;                 // nullable, per column n
    int M, N;
    int relu;                          // epilogue max(0, .) (only without split-K)
    int accumulate;                    // C += (only without split-K)
    int splits;                        // >1: write partial slabs to `slab` instead of C
    float* slab;                       // [splits][M][N] (ld = N)
    int tiles_per_split;               // K tiles (of BK) per split
};

// ---- operand tiles: global -> registers -> LDS ----------------------------------------------------
// Every tile is read through a buffer descriptor (SRD) that covers exactly the valid part of the operand
// window: rows past the matrix edge and K positions past the end of the reduction range fail the hardware
// range check and read as 0.  The K loop therefore has no branches, no address clamps and no per-chunk
// guards; the only per-stage vector work is one `or` per chunk (K-contiguous operands: chunks that START
// past the row end are pushed out of the window) and three selects per chunk (a chunk that STRADDLES the
// row end keeps its valid prefix).  The SRD base / size move with the K position on the scalar unit.
//   (History, profiles/r01/gemm_variants.md: `if (in range) load` serialised the loads behind one
//   s_waitcnt each; clamped addresses + selects cost ~300 VALU per stage; a lean interior path plus a guarded
//   edge path put uniform branches between the MFMAs and kept the schedule from being pinned.)
// Rows past the matrix edge of a row-contiguous operand may read the neighbouring k-row instead of 0: they
// only feed accumulator rows / columns the epilogue never stores.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned SRD_FLAGS = 0x00020000u;      // gfx9 raw buffer, 32-bit data format

constexpr int64_t GEMM_MAX_LD = (int64_t)1 << 22;   // (tile rows - 1) * ld * 4 bytes must fit the 31-bit window offset

// One operand of the block tile: ROWS rows (M or N direction) x BK.  KC: stored [rows][K] (K-contiguous),
// else [K][rows].  Per thread NCH 16-byte chunks:
//   KC : chunk i = (row r0 + 32 i, k = 4 kq),  kq = tid % 8, r0 = tid / 8        LDS image [ROWS][KPAD]
//   !KC: chunk i = (k = kr0 + i KSTEP, rows 4 rq ..), rq = tid % (ROWS/4)         LDS image [BK][ROWS + 4]
template <bool KC, int ROWS>
struct Operand {
    static constexpr int NCH = ROWS * (BK / 4) / 256;
    static constexpr int CPR = ROWS / 4, KSTEP = 256 / CPR;
    static constexpr int LDS_FLOATS = KC ? ROWS * KPAD : BK * (ROWS + 4);
    static_assert(NCH >= 1 && ROWS % 32 == 0, "tile rows");
    int voff[2][NCH];        // byte offset of chunk i inside the window, per K segment (the segments may differ in ld)
    int wofs;                // LDS float offset of chunk 0
    int kpos;                // KC: 4 kq (k offset of this lane's chunks inside a stage)
    f32x4 r[NCH];
    int vc;                  // KC: valid k count of the staged chunks (<= 0 none, >= 4 all)

    __device__ __forceinline__ void init(int64_t ld0, int64_t ld1, int tid) {
        if constexpr (KC) {
            const int kq = tid & 7, r0 = tid >> 3;
            kpos = kq * 4;
            wofs = r0 * KPAD + kq * 4;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                voff[0][i] = (int)(((r0 + 32 * i) * ld0 + kq * 4) * 4);
                voff[1][i] = (int)(((r0 + 32 * i) * ld1 + kq * 4) * 4);
            }
        } else {
            const int rq = tid % CPR, kr0 = tid / CPR;
            kpos = 0;
            wofs = kr0 * (ROWS + 4) + rq * 4;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                voff[0][i] = (int)(((kr0 + i * KSTEP) * ld0 + rq * 4) * 4);
                voff[1][i] = (int)(((kr0 + i * KSTEP) * ld1 + rq * 4) * 4);
            }
        }
        vc = 4;
    }
    static constexpr int lds_chunk_stride() { return KC ? 32 * KPAD : KSTEP * (ROWS + 4); }

    // window of the stage at K position kl of a segment of length kseg: `origin` = &src[row0][0] (KC) or
    // &src[0][row0] (!KC) of that segment, rows_valid rows of the tile exist.  32-bit scalar arithmetic only (the
    // host bounds ld): a `left > 0 ? 64-bit expression : 0` here compiled to scalar BRANCHES inside the K loop.
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t window(const float* origin, int ld, int kl, int kseg,
                                                              int rows_valid) const {
        const int left = kseg - kl;
        const int pos = left > 0 ? 1 : 0;
        if constexpr (KC) {
            const int bytes = ((rows_valid - 1) * ld + left) * 4 * pos;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin + kl), 0, bytes, SRD_FLAGS);
        } else {
            const int krows = left < BK ? left : BK;
            const int bytes = ((krows - 1) * ld + rows_valid) * 4 * pos;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin + (int64_t)kl * ld), 0, bytes, SRD_FLAGS);
        }
    }
    template <int i>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t srd, int seg, int oob) {
        const int vo = seg ? voff[1][i] : voff[0][i];
        r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, KC ? (vo | oob) : vo, 0, 0));
    }
    template <int i>
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
        f32x4 v = r[i];
        if constexpr (KC) {
            v.y = vc > 1 ? v.y : 0.f;
            v.z = vc > 2 ? v.z : 0.f;
            v.w = vc > 3 ? v.w : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + wofs + i * lds_chunk_stride()) = v;
    }
    // fragment of the 32-row block starting at blk_row0 for k-group kg (8 k values): element t feeds MFMA t
    // (k = 8 kg + 4 h + t, h = lane >> 5: any k order is valid as long as A and B agree)
    static __device__ __forceinline__ f32x4 frag(const float* __restrict__ lds, int blk_row0, int kg, int lane) {
        const int i = lane & 31, h = lane >> 5;
        if constexpr (KC) {
            return *reinterpret_cast<const f32x4*>(lds + (blk_row0 + i) * KPAD + kg * 8 + h * 4);
        } else {
            const float* p = lds + (kg * 8 + h * 4) * (ROWS + 4) + blk_row0 + i;
            f32x4 f;
            f.x = p[0]; f.y = p[ROWS + 4]; f.z = p[2 * (ROWS + 4)]; f.w = p[3 * (ROWS + 4)];
            return f;
        }
    }
    static constexpr int frag_reads() { return KC ? 1 : 4; }
};

template <bool AK, bool BKC, int BM, int BN>
constexpr int gemm_lds_bytes() { return 2 * (Operand<AK, BM>::LDS_FLOATS + Operand<BKC, BN>::LDS_FLOATS) * (int)sizeof(float); }

#define GTE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_VALU = 0x002, SG_SALU = 0x004, SG_MFMA = 0x008, SG_VMEM_R = 0x020, SG_DS_R = 0x100, SG_DS_W = 0x200;

// Scheduling pattern of one k-group: NMF MFMAs with NR LDS reads, then NP (ds_write, buffer_load) pairs, spread
// one memory instruction behind every MFMA (two when there are more memory instructions than MFMAs).
template <int NMF, int NR, int NP, bool SCALAR = false>
__device__ __forceinline__ void pin_schedule() {
    constexpr int OPS = NR + 2 * NP;
    constexpr int Q = (OPS + NMF - 1) / NMF;                  // memory instructions per MFMA slot
    // address / window arithmetic (scalar) and the tail selects (vector) ride along in the same slots: left to
    // itself the scheduler clumps ~50 of them between two MFMAs, and a single wave cannot issue the next MFMA
    // until it is through them (the matrix pipe idles once the gap exceeds the 64 cycles of one MFMA)
    constexpr int NV = NP > 0 ? (40 + NMF - 1) / NMF : (SCALAR ? (8 + NMF - 1) / NMF : 0);
    constexpr int NS = SCALAR ? (64 + NMF - 1) / NMF : (NP > 0 ? (8 + NMF - 1) / NMF : 0);
    static_for<NMF>([&](auto Mi) {
        constexpr int m = decltype(Mi)::value;
        GTE_SGB(SG_MFMA, 1);
        if constexpr (NS > 0) GTE_SGB(SG_SALU, NS);
        if constexpr (NV > 0) GTE_SGB(SG_VALU, NV);
        static_for<Q>([&](auto Qi) {
            constexpr int o = m * Q + decltype(Qi)::value;
            if constexpr (o < NR) GTE_SGB(SG_DS_R, 1);
            else if constexpr (o < OPS && ((o - NR) & 1) == 0) GTE_SGB(SG_DS_W, 1);
            else if constexpr (o < OPS) GTE_SGB(SG_VMEM_R, 1);
        });
    });
    __builtin_amdgcn_sched_barrier(0);
}

// AK: A stored [M][K] (K-contiguous) else [K][M].  BKC: B stored [N][K] (K-contiguous) else [K][N].
//
// Software pipeline, one barrier per K stage (LDS double buffer).  While tile t is multiplied out of LDS[t & 1]:
//   k-group 0/1: tile t+1 goes registers -> LDS[(t+1) & 1]; each ds_write is followed by the buffer_load that
//                refills the same registers with tile t+2 (a full stage in flight before it is needed);
//   k-group 2  : only fragment reads; then s_waitcnt lgkmcnt(0) + s_barrier;
//   k-group 3  : its MFMAs hide the first fragment reads of tile t+1.
// Every memory instruction sits behind an MFMA (pin_schedule), so ONE wave per SIMD keeps the matrix pipe fed;
// the loop body is a single basic block.  Measured (profiles/micro/gemm_pipe.hip, M = 24 495): NT K = 1662,
// N = 256: 120 TF (the two-barrier, single-buffer predecessor: 92 TF; vendor hipBLASLt: 98-121 TF);
// N = 1000, K = 2000: 133 TF (predecessor 110, vendor 142).
template <bool AK, bool BKC, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256)
gemm_f32_mfma_kernel(const GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;          // 32x32 tiles per wave
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    using OpA = Operand<AK, BM>;
    using OpB = Operand<BKC, BN>;
    constexpr int A_FLOATS = OpA::LDS_FLOATS, BUF = A_FLOATS + OpB::LDS_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // tile mapping: consecutive logical blocks walk N first (they share the A row panel)
    // N may be split into two column segments fed by two B matrices (dW = dZ^T [h | ahn] in ONE launch)
    const int seg_cols = p.Bn2 ? p.Nseg : p.N;
    const int seg_tiles = (seg_cols + BN - 1) / BN;
    const int tiles_n = p.Bn2 ? 2 * seg_tiles : seg_tiles, tiles_m = (p.M + BM - 1) / BM;
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    // work unit -> (logical tile, K range).  Without a tail split every unit is a whole tile.
    const bool tail = p.sk_pieces > 1 && (int)blockIdx.x >= p.sk_full;
    const int tail_j = tail ? ((int)blockIdx.x - p.sk_full) / p.sk_pieces : 0;
    const int tail_p = tail ? ((int)blockIdx.x - p.sk_full) % p.sk_pieces : 0;
    const unsigned nremap = p.sk_pieces > 1 ? (unsigned)p.sk_full : ntile * (unsigned)(p.splits > 1 ? p.splits : 1);
    // split-K launches (dW: K = the node dimension): units are ordered split-major before the XCD remap, so one XCD
    // works through whole K slabs -- every tile of a split re-reads the same rows of dz / h, and with the splits dealt
    // round-robin over the XCDs each slab was fetched by all eight L2s (PMC: 544 MB per dW launch, 145 MB algorithmic)
    const unsigned lu = tail ? (unsigned)(p.sk_full + tail_j) : gte_xcd_remap(blockIdx.x, nremap);
    const unsigned lb = lu % ntile;
    const int split = (int)(lu / ntile);
    const int tm = lb / tiles_n, tn = lb % tiles_n;
    const int nseg = tn / seg_tiles;                       // 0, or 1 for the second B
    const int m0 = tm * BM, n0 = (tn % seg_tiles) * BN;    // n0: column inside the segment
    const float* Bmat = nseg ? p.Bn2 : p.B;
    const int64_t ldbm = nseg ? p.ldbn2 : p.ldb;

    const int tiles_seg0 = (p.K1 + BK - 1) / BK, tiles_seg1 = (p.K2 + BK - 1) / BK;
    const int total_tiles = tiles_seg0 + tiles_seg1;
    const int t_begin = tail ? tail_p * p.sk_tps : split * p.tiles_per_split;
    const int t_end = min(total_tiles, t_begin + (tail ? p.sk_tps : p.tiles_per_split));

    // operand windows (block-uniform): origin of the tile's rows in each K segment
    const int rowsA = min(BM, p.M - m0), rowsB = min(BN, seg_cols - n0);
    const float* Amat = (nseg && p.An2) ? p.An2 : p.A1;
    const int64_t ldam = (nseg && p.An2) ? p.ldan2 : p.lda1;
    const float* a_org0 = AK ? Amat + (int64_t)m0 * ldam : Amat + m0;
    const float* a_org1 = p.A2 ? (AK ? p.A2 + (int64_t)m0 * p.lda2 : p.A2 + m0) : a_org0;
    const float* b_org0 = BKC ? Bmat + (int64_t)n0 * ldbm : Bmat + n0;
    const float* b_org1 = p.B2 ? (BKC ? p.B2 + (int64_t)n0 * ldbm : p.B2 + n0)
                               : (BKC ? b_org0 + p.K1 : b_org0 + (int64_t)p.K1 * ldbm);

    const int ld_a0 = (int)ldam, ld_a1 = (int)(p.A2 ? p.lda2 : ldam), ld_b = (int)ldbm;   // < GEMM_MAX_LD (host)

    OpA opa;
    OpB opb;
    opa.init(ldam, p.A2 ? p.lda2 : ldam, tid);
    opb.init(ldbm, ldbm, tid);

    constexpr int NCA = OpA::NCH, NCB = OpB::NCH, NC = NCA + NCB, H = (NC + 1) / 2;
    // Cursor over the K tiles this block loads: (segment, k position, tiles left).  `describe()` turns the cursor
    // into the two operand windows + the per-lane tail masks of that tile and advances it; it is pure scalar
    // work plus four vector instructions and runs in k-group 2, one stage before the loads that use it.
    struct TileDesc {
        __amdgpu_buffer_rsrc_t sa, sb;
        int seg, oob_a, oob_b, vca, vcb;
    };
    int c_seg = t_begin >= tiles_seg0 ? 1 : 0;
    int c_kl = (t_begin - (c_seg ? tiles_seg0 : 0)) * BK;
    int c_left = t_end - t_begin;
    // segment lengths pinned in scalar registers (read straight from the kernel argument the compiler turned the
    // selects below into branches around lazy s_load_dword's -- inside the K loop)
    int kseg0 = p.K1, kseg1 = p.K2;
    asm volatile("" : "+s"(kseg0), "+s"(kseg1));
    auto describe = [&]() {
        TileDesc d;
        int kseg = c_seg == 0 ? kseg0 : kseg1;
        kseg = (c_seg < 2 && c_left > 0) ? kseg : 0;
        const int left = kseg - c_kl;
        d.seg = c_seg == 1 ? 1 : 0;
        d.sa = opa.window(d.seg ? a_org1 : a_org0, d.seg ? ld_a1 : ld_a0, c_kl, kseg, rowsA);
        d.sb = opb.window(d.seg ? b_org1 : b_org0, ld_b, c_kl, kseg, rowsB);
        // K-contiguous chunks that start at or past the row end are pushed out of the window (plain arithmetic:
        // a `cond ? x : y` on the offset became a divergent branch around the load)
        d.vca = left - opa.kpos;
        d.vcb = left - opb.kpos;
        d.oob_a = ((d.vca - 1) >> 31) & (int)0x80000000;
        d.oob_b = ((d.vcb - 1) >> 31) & (int)0x80000000;
        c_kl += BK;
        --c_left;
        const bool done = c_kl >= kseg;
        c_seg = done ? (c_seg < 2 ? c_seg + 1 : 2) : c_seg;
        c_kl = done ? 0 : c_kl;
        return d;
    };
    TileDesc D;
    // chunks [FROM, TO) of tile `d` (A chunks first, then B) into the staging registers of (oa, ob)
    auto issue_loads_into = [&](OpA& oa, OpB& ob, const TileDesc& d, auto FROM, auto TO) {
        static_for<NC>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j >= decltype(FROM)::value && j < decltype(TO)::value) {
                if constexpr (j < NCA) oa.template load<j>(d.sa, d.seg, d.oob_a);
                else ob.template load<j - NCA>(d.sb, d.seg, d.oob_b);
            }
        });
    };
    auto issue_loads = [&](auto FROM, auto TO) { issue_loads_into(opa, opb, D, FROM, TO); };
    auto staged = [&]() { opa.vc = D.vca; opb.vc = D.vcb; };   // the described tile now sits in the staging registers
    auto store_chunks_from = [&](const OpA& oa, const OpB& ob, int buf, auto FROM, auto TO) {
        float* la = lds + buf * BUF;
        float* lbp = la + A_FLOATS;
        static_for<NC>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j >= decltype(FROM)::value && j < decltype(TO)::value) {
                if constexpr (j < NCA) oa.template store<j>(la);
                else ob.template store<j - NCA>(lbp);
            }
        });
    };
    auto store_chunks = [&](int buf, auto FROM, auto TO) { store_chunks_from(opa, opb, buf, FROM, TO); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int buf, int kg, int slot) {
        const float* la = lds + buf * BUF;
        const float* lbp = la + A_FLOATS;
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[slot][a] = OpA::frag(la, (wm * TM + a) * 32, kg, lane);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[slot][b] = OpB::frag(lbp, (wn * TN + b) * 32, kg, lane);
    };
    auto mfma_group = [&](int slot) {                      // b fastest: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][tt], fb[slot][b][tt], acc[a][b], 0, 0, 0);
    };
    constexpr int NMF = 4 * TM * TN;                       // MFMAs per k-group
    constexpr int NR = TM * OpA::frag_reads() + TN * OpB::frag_reads();
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, H>;
    using IN = std::integral_constant<int, NC>;

    if (t_begin < t_end) {
        // prologue: tile t_begin -> LDS[0], tile t_begin + 1 -> staging registers.  BOTH tiles are requested before
        // the first wait (tile t_begin goes through a second, prologue-only register set): one memory round trip
        // before the first MFMA instead of two
        {
            OpA pa = opa;
            OpB pb = opb;
            const TileDesc D0 = describe();
            issue_loads_into(pa, pb, D0, I0{}, IN{});
            D = describe();
            issue_loads(I0{}, IN{});
            pa.vc = D0.vca; pb.vc = D0.vcb;
            store_chunks_from(pa, pb, 0, I0{}, IN{});
            staged();
            D = describe();
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(0, 0, 0);
        int cur = 0;
        for (int t = t_begin; t < t_end; ++t, cur ^= 1) {
            read_frags(cur, 1, 1);
            store_chunks(cur ^ 1, I0{}, IH{});
            issue_loads(I0{}, IH{});
            mfma_group(0);
            pin_schedule<NMF, NR, H>();
            read_frags(cur, 2, 0);
            store_chunks(cur ^ 1, IH{}, IN{});
            issue_loads(IH{}, IN{});
            mfma_group(1);
            pin_schedule<NMF, NR, NC - H>();
            staged();
            read_frags(cur, 3, 1);
            D = describe();
            mfma_group(0);
            pin_schedule<NMF, NR, 0, true>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_frags(cur ^ 1, 0, 0);
            mfma_group(1);
            pin_schedule<NMF, NR, 0>();
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    if (tail) {                                            // tile-local partial, finished by gemm_tail_fixup_kernel
        float* sp = p.sk_slab + (int64_t)(tail_j * p.sk_pieces + tail_p) * (BM * BN);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sp[((wm * TM + a) * 32 + hrow + (r & 3) + 8 * (r >> 2)) * BN + (wn * TN + b) * 32 + col_l] = acc[a][b][r];
        return;
    }
    float* outp = p.splits > 1 ? p.slab + (int64_t)split * p.M * p.N : p.C;
    const int64_t ldo = p.splits > 1 ? p.N : p.ldc;
    // Branch-free stores through a buffer descriptor over the output (round 2): rows past M and columns past the tile's segment
    // get an offset outside the window and are dropped by the range check -- no per-element compare / exec mask / 64-bit
    // address.  The guarded pointer loop below runs ~11 vector instructions and two branches per element (64 elements per
    // wave), and its stores trickle out behind them at the end of every tile; measured on the split kernel: layer-0 forward
    // 141 -> 128 us, layer-1 forward 57 -> 50 us.  Needs every offset of the padded tile grid to fit 31 bits.
    const int64_t extent = ((int64_t)p.M + BM) * ldo * 4;
    if (extent < ((int64_t)1 << 31)) {
        const bool post = p.splits <= 1 && (p.accumulate || p.relu);
        const int ncols_out = p.splits > 1 ? p.N : (p.Bn2 ? 2 * seg_cols : p.N);
        const __amdgpu_buffer_rsrc_t c_srd =
            __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(((int64_t)(p.M - 1) * ldo + ncols_out) * 4), SRD_FLAGS);
        const int ld4 = (int)ldo * 4;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col_in_seg = n0 + (wn * TN + b) * 32 + col_l;
            const int col = nseg * seg_cols + col_in_seg;
            const int coff = col_in_seg < seg_cols ? col * 4 : (int)0x80000000;          // out of the window
            const float bv = (p.bias && p.splits <= 1 && col_in_seg < seg_cols && (p.bias_cols <= 0 || col < p.bias_cols)) ? p.bias[col] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int roff0 = (m0 + (wm * TM + a) * 32 + hrow) * ld4 + coff;
                if (!post) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // through a VGPR on purpose: with an accumulator element as the store's data operand hipcc 7.2 emitted
                        // element 0 for all sixteen stores (gemm_bf16.hip); the add alone is not a guarantee
                        float v = acc[a][b][r] + bv;
                        asm volatile("" : "+v"(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_srd,
                                                              roff0 + ((r & 3) + 8 * (r >> 2)) * ld4, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = roff0 + ((r & 3) + 8 * (r >> 2)) * ld4;
                        float v = acc[a][b][r] + bv;
                        if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_srd, off, 0, 0));
                        if (p.relu) v = fmaxf(v, 0.f);
                        asm volatile("" : "+v"(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_srd, off, 0, 0);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col_in_seg = n0 + (wn * TN + b) * 32 + col_l;
        if (col_in_seg >= seg_cols) continue;
        const int col = nseg * seg_cols + col_in_seg;
        const float bv = (p.bias && p.splits <= 1 && (p.bias_cols <= 0 || col < p.bias_cols)) ? p.bias[col] : 0.f;
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

#include "gemm_split.h"

// Finishes the tail tiles of a split launch: C tile = bias + sum over the pieces (piece order: deterministic), with the
// same bias / accumulate / relu rules as the GEMM epilogue.  grid = (tail tiles, BM * BN / 1024), 4 elements per thread.
template <int BM, int BN>
__global__ void __launch_bounds__(256)
gemm_tail_fixup_kernel(const GemmParams p) {
    const int seg_cols = p.Bn2 ? p.Nseg : p.N;
    const int seg_tiles = (seg_cols + BN - 1) / BN;
    const int tiles_n = p.Bn2 ? 2 * seg_tiles : seg_tiles;
    const int lb = p.sk_full + (int)blockIdx.x;
    const int tm = lb / tiles_n, tn = lb % tiles_n, nseg = tn / seg_tiles;
    const int m0 = tm * BM, n0 = (tn % seg_tiles) * BN;
    const float* sp = p.sk_slab + (int64_t)blockIdx.x * p.sk_pieces * (BM * BN);
    const int e = ((int)blockIdx.y * 256 + (int)threadIdx.x) * 4;
    const int r = e / BN, c = e % BN;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pc = 0; pc < p.sk_pieces; ++pc) {
        const float4 t = *reinterpret_cast<const float4*>(sp + (int64_t)pc * (BM * BN) + e);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int row = m0 + r;
    if (row >= p.M) return;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col_in_seg = n0 + c + q;
        if (col_in_seg >= seg_cols) continue;
        const int col = nseg * seg_cols + col_in_seg;
        float o = vv[q] + ((p.bias && (p.bias_cols <= 0 || col < p.bias_cols)) ? p.bias[col] : 0.f);
        float* dst = p.C + (int64_t)row * p.ldc + col;
        if (p.accumulate) o += *dst;
        if (p.relu) o = fmaxf(o, 0.f);
        *dst = o;
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
                     const float* __restrict__ bias, int accumulate, int relu, int bias_cols = 0) {
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
        if (bias && (bias_cols <= 0 || n < bias_cols)) s += bias[n];
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
    const int nn = min(n, N - 1);
    __syncthreads();
    float acc[SK_MAX];
#pragma unroll
    for (int i = 0; i < SK_MAX; ++i) acc[i] = 0.f;
    // Eight k rows per trip of a loop that is NOT unrolled: fully unrolled (32 rows), the compiler hoisted all 512 broadcast
    // reads of sA in front of the FMAs -- 256 VGPRs + 256 AGPRs and 148 bytes of scratch per lane.  Eight B loads in flight per
    // thread, the other waves of the CU cover the rest.
#pragma unroll 1
    for (int r0 = 0; r0 < SK_ROWS; r0 += 8) {
        float b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) b[r] = B[(int64_t)(k0 + min(r0 + r, nk - 1)) * ldb + nn];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < SK_MAX; ++i) acc[i] = fmaf(sA[r0 + r][i], b[r], acc[i]);   // rows >= nk hold zeros in sA
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
    const bool n2 = p.Bn2 && n >= p.Nseg;
    for (int seg = 0; seg < 2; ++seg) {
        const float* A = seg ? p.A2 : ((n2 && p.An2) ? p.An2 : p.A1);
        const int64_t lda = seg ? p.lda2 : ((n2 && p.An2) ? p.ldan2 : p.lda1);
        const int K = seg ? p.K2 : p.K1, kb = (seg && !p.B2) ? p.K1 : 0;
        for (int k = 0; k < K; ++k) {
            const float a = AK ? A[(int64_t)m * lda + k] : A[(int64_t)k * lda + m];
            const float* Bm = (seg && p.B2) ? p.B2 : (n2 ? p.Bn2 : p.B);
            const int64_t ldbm = n2 ? p.ldbn2 : p.ldb;
            const int nn = n2 ? n - p.Nseg : n;
            const float b = BKC ? Bm[(int64_t)nn * ldbm + kb + k] : Bm[(int64_t)(kb + k) * ldbm + nn];
            acc = fmaf(a, b, acc);
        }
    }
    if (p.bias && (p.bias_cols <= 0 || n < p.bias_cols)) acc += p.bias[n];
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

// GEMM arithmetic mode (gte_gemm_set_mode): process-wide -- autograd runs the backward GEMMs on another host thread.
// -1 = not yet read from the environment (GTE_GEMM_MODE=f32 selects the fp32 MFMA kernel, anything else the split mode).
std::atomic<int> g_gemm_mode{-1};
// per-thread override (gte_gemm_set_thread_mode): -1 = none.  Lets one caller (the GAT's bf16 configuration) run ITS GEMMs in
// the split mode without changing what other threads / modules get.
thread_local int t_gemm_mode = -1;
int gemm_mode() {
    if (t_gemm_mode >= 0) return t_gemm_mode;
    int m = g_gemm_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        // default since round 3: the split mode (three exact bf16 pieces per operand: not narrower than fp32, error against fp64
        // below the fp32 MFMA kernel's, every reference golden test green in it); GTE_GEMM_MODE=f32 selects the fp32 MFMA kernel
        const char* e = getenv("GTE_GEMM_MODE");
        m = (e && (!strcmp(e, "f32") || !strcmp(e, "fp32") || !strcmp(e, "0"))) ? GTE_GEMM_F32 : GTE_GEMM_SPLIT_BF16;
        g_gemm_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

struct Plan { int bm, bn, tiles, splits, tiles_per_split, splits_bound; };

Plan make_plan(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t Nseg = 0) {
    Plan pl;
    const int cus = gte::device_props().cus;
    if (N <= 32) { pl.bm = 128; pl.bn = 32; }
    else if (M <= 32) { pl.bm = 32; pl.bn = 128; }
    else {
        // Tile choice by a measured cost model.  T tiles run ceil(T / CUs) rounds of one tile per CU (two co-resident
        // workgroups share a CU's matrix pipe, so rounds count CUs, not slots); a round costs F + s * K microseconds
        // (linear fits over K at 768 tiles, profiles/r01/gemm_variants.md): 128x128: 8.07 + 0.0594 K, 64x128:
        // 2.67 + 0.0335 K -- the big tile is 13 % cheaper per flop but pays three times the fixed cost (dispatch, prologue,
        // the burst of C stores at the end of a round), and quantises coarser.  With the tail split available
        // (plan_tail) a last round of r <= CUs/2 tiles costs one round over K/S plus the fix-up launch (~7 us).
        const int ktiles_all = (int)(gte::ceil_div(K1, BK) + gte::ceil_div(K2, BK));
        const double Ktot = (double)(K1 + K2);
        const bool tail_ok = gte::tail_workspace().ptr != nullptr;
        auto cost = [&](int64_t T, double F, double sl) -> double {
            const double round = F + sl * Ktot;
            if (T <= cus) return round;
            const int64_t full = T / cus, r = T % cus;
            if (r == 0) return (double)full * round;
            if (tail_ok && 2 * r <= cus) {
                int64_t S = cus / r;
                if (S > ktiles_all / 4) S = ktiles_all / 4;
                if (S >= 2) return (double)full * round + F + sl * Ktot / (double)S + 7.0;
            }
            return (double)(full + 1) * round;
        };
        const int64_t ncol = Nseg > 0 ? 2 * gte::ceil_div(Nseg, 128) : gte::ceil_div(N, 128);
        const int64_t t128 = gte::ceil_div(M, 128) * ncol, t64 = gte::ceil_div(M, 64) * ncol;
        pl.bn = 128;
        // Re-fitted after the branch-free store epilogue (round 2, profiles/debug/gemm_split_tiles.py, 512 / 1024 tiles over K =
        // 128 .. 1662): fp32 MFMA 128x128: 4.7 + 0.0589 K, 64x128: 2.5 + 0.0315 K (the fixed cost of the big tile was 8.07);
        // split mode (gemm_split.h) 128x128: 3.8 + 0.0458 K, 64x128: 2.2 + 0.0273 K.
        const bool split = gemm_mode() == GTE_GEMM_SPLIT_BF16;
        const double c64 = split ? cost(t64, 2.2, 0.0273) : cost(t64, 2.5, 0.0315);
        const double c128 = split ? cost(t128, 3.8, 0.0458) : cost(t128, 4.7, 0.0589);
        pl.bm = (t128 >= cus && c64 < c128) ? 64 : 128;
        // split-K with very few output tiles (dW of a 256 x 256 layer: 4 tiles over 24 k nodes): smaller tiles ->
        // half the K splits -> half the slab bytes (measured 58 -> 44 us; at 14 tiles, 256 x 831, it loses: 147 -> 161)
        if (t128 <= 8 && M >= 64) pl.bm = 64;
        static const int force_bm = GTE_MEASURE_INT("GTE_GEMM_BM", 0);
        if (force_bm == 64 || force_bm == 128) pl.bm = force_bm;
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
    // pl.splits <= splits, and `splits` never shrinks when K grows (the tile shape of a split-K plan does not depend on
    // K): the bound a workspace query may use for "any K up to this one".  pl.splits itself is NOT monotonic in K
    // (896 K tiles -> 35 splits, 846 -> 36): sizing a workspace from the capacity's exact plan came out 2.8 % short.
    pl.splits_bound = splits;
    return pl;
}

// Tail split of a whole-K launch.  A launch of T tiles on the chip runs ceil(T / CUs) rounds of "one tile per CU at
// full rate" (two co-resident workgroups share a CU's matrix pipe, so the count is per CU, not per slot).  When the last
// round holds r <= CUs/2 tiles, (CUs - r) CUs idle for a whole tile time: 800 tiles = 3.125 rounds cost 4.  The r tail
// tiles are instead cut into S = CUs / r K ranges (>= 4 K tiles each): r * S <= CUs short units that fill the round
// evenly, tile-local partials in the caller's tail workspace, one small fix-up launch.  (A full stream-K decomposition
// would route nearly every tile of these 3-round launches through partials: ~3x the C traffic.)
template <int BM, int BN>
bool plan_tail(GemmParams& p, int tiles, dim3& grid) {
    p.sk_full = tiles; p.sk_pieces = 1; p.sk_tps = 0; p.sk_slab = nullptr;
    const gte::TailWorkspace tw = gte::tail_workspace();
    const int cus = gte::device_props().cus;
    if (!tw.ptr || p.splits > 1 || tiles < cus) return false;
    const int r = tiles % cus;
    if (r == 0 || 2 * r > cus) return false;
    const int ktiles = (int)(gte::ceil_div(p.K1, BK) + gte::ceil_div(p.K2, BK));
    int S = cus / r;
    if (S > ktiles / 4) S = ktiles / 4;
    if (S < 2) return false;
    const int tps = (int)gte::ceil_div(ktiles, S);
    S = (int)gte::ceil_div(ktiles, tps);
    if (S < 2 || (int64_t)r * S * BM * BN * 4 > tw.bytes) return false;
    p.sk_full = tiles - r; p.sk_pieces = S; p.sk_tps = tps; p.sk_slab = tw.ptr;
    grid = dim3((unsigned)(p.sk_full + r * S), 1);
    return true;
}

template <bool AK, bool BKC, int BM, int BN, int WM, int WN>
void launch_tile(GemmParams p, dim3 grid, hipStream_t s) {
    constexpr int shm = gemm_lds_bytes<AK, BKC, BM, BN>();
    static bool configured = false;                       // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<AK, BKC, BM, BN, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        configured = true;
    }
    const int tiles = (int)grid.x;
    const bool tail = plan_tail<BM, BN>(p, tiles, grid);
    if constexpr (BN == 128 && (BM == 128 || BM == 64) && WM == 2 && WN == 2) {
        if (gemm_mode() == GTE_GEMM_SPLIT_BF16) {
            constexpr int shm_s = gemm_split_lds_bytes<AK, BKC, BM, BN>();
            static bool configured_s = false;
            if (!configured_s) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<AK, BKC, BM, BN, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, shm_s);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<AK, BKC, BM, BN, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, shm_s);
                configured_s = true;
            }
            // K-contiguous operands with a K range that is not a multiple of 4 need the chunk tail masks
            const bool ktail = (AK || BKC) && (p.K1 % 4 != 0 || p.K2 % 4 != 0);
            if (ktail) hipLaunchKernelGGL((gemm_split_kernel<AK, BKC, BM, BN, true>), grid, dim3(256), shm_s, s, p);
            else hipLaunchKernelGGL((gemm_split_kernel<AK, BKC, BM, BN, false>), grid, dim3(256), shm_s, s, p);
            if (tail)
                hipLaunchKernelGGL((gemm_tail_fixup_kernel<BM, BN>), dim3((unsigned)(tiles - p.sk_full), BM * BN / 1024), dim3(256), 0, s, p);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, BM, BN, WM, WN>), grid, dim3(256), shm, s, p);
    if (tail)
        hipLaunchKernelGGL((gemm_tail_fixup_kernel<BM, BN>), dim3((unsigned)(tiles - p.sk_full), BM * BN / 1024), dim3(256), 0, s, p);
}

template <bool AK, bool BKC>
int launch_shape(const GemmParams& p, const Plan& pl, hipStream_t s) {
    dim3 grid((unsigned)(pl.tiles * pl.splits), 1);
    if (pl.bm == 128 && pl.bn == 128) launch_tile<AK, BKC, 128, 128, 2, 2>(p, grid, s);
    else if (pl.bm == 64 && pl.bn == 128) launch_tile<AK, BKC, 64, 128, 2, 2>(p, grid, s);
    else if (pl.bn == 32) launch_tile<AK, BKC, 128, 32, 4, 1>(p, grid, s);
    else launch_tile<AK, BKC, 32, 128, 1, 4>(p, grid, s);
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

// may_defer: only the weight-gradient entry points (gte_sage_linear_dw, gte_sage_qform_dw) hand their split-K fold to an
// open deferral (gte_fold_defer_begin): nothing reads dW before the flush.  Every other caller -- gte_gemm_f32 in
// particular, whose result (dh) the next kernel of a backward reads -- folds immediately.
int run_gemm(bool ak, bool bkc, GemmParams p, void* workspace, int64_t workspace_bytes, hipStream_t s, bool may_defer = false) {
    if (p.M == 0 || p.N == 0) return GTE_OK;
    if (needs_small_path(ak, bkc, p)) return run_small(ak, bkc, p, s);
    if (p.lda1 >= GEMM_MAX_LD || p.lda2 >= GEMM_MAX_LD || p.ldb >= GEMM_MAX_LD || p.ldbn2 >= GEMM_MAX_LD || p.ldan2 >= GEMM_MAX_LD)
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_f32: leading dimensions must be < 2^22 elements");
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
        if (may_defer && !p.bias && !accumulate && !relu && gte::defer_fold(p.slab, mn, pl.splits, p.M, p.N, p.C, p.ldc)) return GTE_OK;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gte::ceil_div(mn, RED_E)), dim3(RED_E * RED_S), 0, s, p.slab,
                           pl.splits, mn, p.N, p.C, p.ldc, p.bias, accumulate, relu, p.bias_cols);
        return gte::check_launch("gemm_f32 split-K reduce");
    }
    return GTE_OK;
}

int64_t gemm_workspace(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t Nseg = 0) {
    if (M <= 0 || N <= 0) return 256;
    const Plan pl = make_plan(M, N, K1, K2, Nseg);
    return pl.splits_bound > 1 ? gte::round_up((int64_t)pl.splits_bound * M * N * 4, 256) : 256;
}

// ------------------------------- LayerNorm + ReLU, forward ----------------------------------------
// y = relu?( gamma * (z - mean) * rstd + beta ); one wave per row, two-pass mean/variance (biased
// variance, eps inside the sqrt: torch.nn.LayerNorm).  In place (y == z) is allowed.
constexpr int LN_CACHE = 16;      // elements per lane kept in registers -> rows up to 1024 wide

__device__ __forceinline__ float wave_sum(float v) { return gte_group_sum<64>(v); }

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

// 16-byte accesses, two rows in flight per wave: lane l owns columns 4 (l + 64 v) .. + 3 (n % 4 == 0, n <= 256 NV).
// Same two-pass statistics.  (The scalar version above moves 4 bytes per lane per instruction: 13.9 us for 50 MB, a plain
// device copy of the same bytes takes 8.3 us on this chip.)
template <int NV>
__global__ void __launch_bounds__(256)
ln_relu_fwd_vec_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ gamma,
                       const float* __restrict__ beta, float eps, int relu, float* __restrict__ y, int64_t ldy,
                       float* __restrict__ stats, int M, int n) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= M) return;
    float g[NV][4], b[NV][4];
    bool okv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int j = 4 * (lane + 64 * v);
        okv[v] = j < n;
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[v][e] = okv[v] ? gamma[j + e] : 0.f; b[v][e] = okv[v] ? beta[j + e] : 0.f; }
    }
    float c[2][NV][4];
    bool rok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        rok[u] = row0 + u < M;
        const int r = rok[u] ? row0 + u : row0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f4u t{0.f, 0.f, 0.f, 0.f};
            if (okv[v]) t = *reinterpret_cast<const f4u*>(z + (int64_t)r * ldz + 4 * (lane + 64 * v));
            c[u][v][0] = t.x; c[u][v][1] = t.y; c[u][v][2] = t.z; c[u][v][3] = t.w;
        }
    }
    const float inv_n = 1.0f / (float)n;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!rok[u]) continue;                             // wave-uniform
        const int r = row0 + u;
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += c[u][v][e];
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = okv[v] ? c[u][v][e] - mean : 0.f; q = fmaf(d, d, q); }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        if (stats && lane == 0) { stats[r] = mean; stats[M + r] = rstd; }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (okv[v]) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = ln_affine((c[u][v][e] - mean) * rstd, g[v][e], b[v][e]);
                    if (relu) o[e] = fmaxf(o[e], 0.f);
                }
                f4u t; t.x = o[0]; t.y = o[1]; t.z = o[2]; t.w = o[3];
                *reinterpret_cast<f4u*>(y + (int64_t)r * ldy + 4 * (lane + 64 * v)) = t;
            }
        }
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

// Same maths with 16-byte accesses: lane l owns columns 4 (l + 64 v) .. + 3, v < NV (n % 4 == 0, n <= 256 NV).  The scalar
// version above moves 4 bytes per lane per instruction (12 memory instructions per 256-wide row): 30 us for the 75 MB
// of a 24 k x 256 layer, 2.5 TB/s; two rows are in flight per wave here.
template <int NV>
__global__ void __launch_bounds__(256)
ln_relu_bwd_vec_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z, int64_t ldz,
                       const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                       int relu, float* __restrict__ dz, int64_t lddz, float* __restrict__ partial, int M, int n,
                       char* __restrict__ dzp3 = nullptr, int64_t ldp3 = 0) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][3][NV*256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool has_ln = gamma != nullptr;
    float gam[NV][4], bet[NV][4], s_dg[NV][4], s_db[NV][4], s_dbias[NV][4];
    bool okv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int j = 4 * (lane + 64 * v);
        okv[v] = j < n;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = (has_ln && okv[v]) ? gamma[j + e] : 1.f;
            bet[v][e] = (has_ln && okv[v]) ? beta[j + e] : 0.f;
            s_dg[v][e] = s_db[v][e] = s_dbias[v][e] = 0.f;
        }
    }
    const float inv_n = 1.0f / (float)n;
    const int stride = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + wave; row < M; row += 2 * stride) {
        // two rows per trip: all four 16-byte loads are issued before the first reduction
        float gy[2][NV][4], zz[2][NV][4];
        float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
        bool rok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = row + u * stride;
            rok[u] = r < M;
            const int rc = rok[u] ? r : row;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int j = 4 * (lane + 64 * v);
                f4u a{0.f, 0.f, 0.f, 0.f}, b{0.f, 0.f, 0.f, 0.f};
                if (okv[v]) {
                    a = *reinterpret_cast<const f4u*>(dy + (int64_t)rc * lddy + j);
                    b = *reinterpret_cast<const f4u*>(z + (int64_t)rc * ldz + j);
                }
                gy[u][v][0] = a.x; gy[u][v][1] = a.y; gy[u][v][2] = a.z; gy[u][v][3] = a.w;
                zz[u][v][0] = b.x; zz[u][v][1] = b.y; zz[u][v][2] = b.z; zz[u][v][3] = b.w;
            }
            if (has_ln) { mean[u] = stats[rc]; rstd[u] = stats[M + rc]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!rok[u]) continue;                       // wave-uniform
            float* dzr = dz + (int64_t)(row + u * stride) * lddz;
            if (has_ln) {
                float xh[NV][4], g[NV][4];
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)                 // (gte_common.h: the arithmetic shared with the GEMM epilogue form)
                    gte_ln_bwd_pre4(gy[u][v], zz[u][v], mean[u], rstd[u], gam[v], bet[v], okv[v], relu, xh[v], g[v], a, b);
                const float c1 = wave_sum(a) * inv_n, c2 = wave_sum(b) * inv_n;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float d[4];
                    gte_ln_bwd_post4(g[v], xh[v], gam[v], rstd[u], c1, c2, okv[v], d, s_dg[v], s_db[v], s_dbias[v]);
                    if (okv[v]) {
                        f4u o; o.x = d[0]; o.y = d[1]; o.z = d[2]; o.w = d[3];
                        *reinterpret_cast<f4u*>(dzr + 4 * (lane + 64 * v)) = o;
                        // ... and as a P3 image: dz is the A operand of the dX and dW planes GEMMs
                        if (dzp3) p3::store4(dzp3 + (int64_t)(row + u * stride) * ldp3, 4 * (lane + 64 * v), d[0], d[1], d[2], d[3]);
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d[e] = (relu && zz[u][v][e] <= 0.f) ? 0.f : gy[u][v][e];
                        s_dbias[v][e] += d[e];
                    }
                    if (okv[v]) {
                        f4u o; o.x = d[0]; o.y = d[1]; o.z = d[2]; o.w = d[3];
                        *reinterpret_cast<f4u*>(dzr + 4 * (lane + 64 * v)) = o;
                        if (dzp3) p3::store4(dzp3 + (int64_t)(row + u * stride) * ldp3, 4 * (lane + 64 * v), d[0], d[1], d[2], d[3]);
                    }
                }
            }
        }
    }
    constexpr int W = NV * 256;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * (lane + 64 * v) + e;
            red[(wave * 3 + 0) * W + j] = s_dg[v][e];
            red[(wave * 3 + 1) * W + j] = s_db[v][e];
            red[(wave * 3 + 2) * W + j] = s_dbias[v][e];
        }
    __syncthreads();
    float* pp = partial + (int64_t)blockIdx.x * 3 * n;
    for (int i = threadIdx.x; i < 3 * W; i += 256) {
        const int q = i / W, j = i - q * W;
        if (j < n) pp[q * n + j] = red[(0 * 3 + q) * W + j] + red[(1 * 3 + q) * W + j] + red[(2 * 3 + q) * W + j] +
                                   red[(3 * 3 + q) * W + j];
    }
}

// ---- any width up to 1024 (the reference's own run shapes: hidden 1000, or int(calculate_hidden) = 96 ... 218) ----------------
// Rows PADDED to a multiple of 4 floats (ld >= n rounded up to 4): lane l owns columns 4 (l + 64 v) .. + 3, v < NV; per-element
// masks instead of the per-chunk mask of the vec kernels; columns n .. up to the next multiple of 4 of the fp32 outputs and up to
// the next multiple of 16 of the P3 images are written as zeros.  Same two-pass statistics / the same backward arithmetic.
template <int NV, int RF>
__global__ void __launch_bounds__(256)
ln_relu_fwd_gen_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ gamma, const float* __restrict__ beta,
                       float eps, int relu, float* __restrict__ y, int64_t ldy, char* __restrict__ yp3, int64_t ldyp3,
                       float* __restrict__ stats, int M, int n) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RF;
    if (row0 >= M) return;
    const int n16 = (n + 15) & ~15;
    float g[NV][4], b[NV][4];
    bool okv[NV], imv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int j = 4 * (lane + 64 * v);
        okv[v] = j < n;
        imv[v] = j < n16;
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[v][e] = j + e < n ? gamma[j + e] : 0.f; b[v][e] = j + e < n ? beta[j + e] : 0.f; }
    }
    float c[RF][NV][4];
    bool rok[RF];
#pragma unroll
    for (int u = 0; u < RF; ++u) {
        rok[u] = row0 + u < M;
        const int r = rok[u] ? row0 + u : row0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int j = 4 * (lane + 64 * v);
            f4u t{0.f, 0.f, 0.f, 0.f};
            if (okv[v]) t = *reinterpret_cast<const f4u*>(z + (int64_t)r * ldz + j);
            c[u][v][0] = t.x; c[u][v][1] = j + 1 < n ? t.y : 0.f; c[u][v][2] = j + 2 < n ? t.z : 0.f; c[u][v][3] = j + 3 < n ? t.w : 0.f;
        }
    }
    const float inv_n = 1.0f / (float)n;
#pragma unroll
    for (int u = 0; u < RF; ++u) {
        if (!rok[u]) continue;                             // wave-uniform
        const int r = row0 + u;
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += c[u][v][e];
        // (z - mean as ONE fused multiply-add of the row sum, written out: hipcc contracts `c - wave_sum(s) * inv_n` this way, and
        // the LayerNorm-forward epilogue of the planes GEMM -- gemm_p3.hip, LNB == 4 -- is held bit for bit to this kernel)
        const float wsum = wave_sum(s);
        const float mean = wsum * inv_n;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = 4 * (lane + 64 * v) + e < n ? fmaf(-inv_n, wsum, c[u][v][e]) : 0.f; q = fmaf(d, d, q); }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        if (stats && lane == 0) { stats[r] = mean; stats[M + r] = rstd; }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!imv[v]) continue;
            const int j = 4 * (lane + 64 * v);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = ln_affine(fmaf(-inv_n, wsum, c[u][v][e]) * rstd, g[v][e], b[v][e]);
                if (relu) o[e] = fmaxf(o[e], 0.f);
                if (j + e >= n) o[e] = 0.f;
            }
            if (y && okv[v]) {
                f4u t; t.x = o[0]; t.y = o[1]; t.z = o[2]; t.w = o[3];
                *reinterpret_cast<f4u*>(y + (int64_t)r * ldy + j) = t;
            }
            if (yp3) p3::store4(yp3 + (int64_t)r * ldyp3, j, o[0], o[1], o[2], o[3]);
        }
    }
}

template <int NV, int RF>
__global__ void __launch_bounds__(256)
ln_relu_bwd_gen_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z, int64_t ldz,
                       const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                       int relu, float* __restrict__ dz, int64_t lddz, float* __restrict__ partial, int M, int n,
                       char* __restrict__ dzp3, int64_t ldp3) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][3][NV*256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = (n + 15) & ~15;
    float gam[NV][4], bet[NV][4], s_dg[NV][4], s_db[NV][4], s_dbias[NV][4];
    bool okv[NV], imv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int j = 4 * (lane + 64 * v);
        okv[v] = j < n;
        imv[v] = j < n16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gam[v][e] = j + e < n ? gamma[j + e] : 1.f;
            bet[v][e] = j + e < n ? beta[j + e] : 0.f;
            s_dg[v][e] = s_db[v][e] = s_dbias[v][e] = 0.f;
        }
    }
    const float inv_n = 1.0f / (float)n;
    const int stride = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + wave; row < M; row += RF * stride) {
        float gy[RF][NV][4], zz[RF][NV][4];
        float mean[RF], rstd[RF];
        bool rok[RF];
#pragma unroll
        for (int u = 0; u < RF; ++u) {
            const int r = row + u * stride;
            rok[u] = r < M;
            const int rc = rok[u] ? r : row;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int j = 4 * (lane + 64 * v);
                f4u a{0.f, 0.f, 0.f, 0.f}, b{0.f, 0.f, 0.f, 0.f};
                if (okv[v]) {
                    a = *reinterpret_cast<const f4u*>(dy + (int64_t)rc * lddy + j);
                    b = *reinterpret_cast<const f4u*>(z + (int64_t)rc * ldz + j);
                }
                gy[u][v][0] = a.x; gy[u][v][1] = a.y; gy[u][v][2] = a.z; gy[u][v][3] = a.w;
                zz[u][v][0] = b.x; zz[u][v][1] = b.y; zz[u][v][2] = b.z; zz[u][v][3] = b.w;
            }
            mean[u] = stats[rc]; rstd[u] = stats[M + rc];
        }
#pragma unroll
        for (int u = 0; u < RF; ++u) {
            if (!rok[u]) continue;                       // wave-uniform
            const int64_t r = row + u * stride;
            float xh[NV][4], g[NV][4];
            float a = 0.f, b = 0.f;
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = 4 * (lane + 64 * v) + e < n;
                        xh[v][e] = ok ? (zz[u][v][e] - mean[u]) * rstd[u] : 0.f;
                        float gv = ok ? gy[u][v][e] : 0.f;
                        if (relu && fmaf(xh[v][e], gam[v][e], bet[v][e]) <= 0.f) gv = 0.f;
                        g[v][e] = gv;
                        const float dxh = gv * gam[v][e];
                        a = a + dxh;
                        b = fmaf(dxh, xh[v][e], b);
                    }
            }
            const float c1 = wave_sum(a) * inv_n, c2 = wave_sum(b) * inv_n;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (!imv[v]) continue;
                const int j = 4 * (lane + 64 * v);
                float d[4];
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t0 = g[v][e] * gam[v][e];
                        const float t1 = xh[v][e] * c2;
                        d[e] = j + e < n ? rstd[u] * ((t0 - c1) - t1) : 0.f;
                        s_dg[v][e] = fmaf(g[v][e], xh[v][e], s_dg[v][e]);
                        s_db[v][e] = s_db[v][e] + g[v][e];
                        s_dbias[v][e] = s_dbias[v][e] + d[e];
                    }
                }
                if (okv[v]) {
                    f4u o; o.x = d[0]; o.y = d[1]; o.z = d[2]; o.w = d[3];
                    *reinterpret_cast<f4u*>(dz + r * lddz + j) = o;
                }
                if (dzp3) p3::store4(dzp3 + r * ldp3, j, d[0], d[1], d[2], d[3]);
            }
        }
    }
    constexpr int W = NV * 256;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * (lane + 64 * v) + e;
            red[(wave * 3 + 0) * W + j] = s_dg[v][e];
            red[(wave * 3 + 1) * W + j] = s_db[v][e];
            red[(wave * 3 + 2) * W + j] = s_dbias[v][e];
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
    static const int forced = GTE_MEASURE_INT("GTE_LNB_BLOCKS", 0);
    const int cap = forced > 0 ? forced : LNB_MAX_BLOCKS;
    const int64_t b = gte::ceil_div(M, 4);
    return (int)(b < cap ? b : cap);
}

// ---- short-K layer (BBOX features: 13 + 13 inputs): linear + bias + LayerNorm + ReLU in ONE pass ------------------------
// z[r, :] = a1[r, 0:k1] W[:, 0:k1]^T + a2[r, 0:k2] W[:, k1:K]^T + bias,   K = k1 + k2 <= 64, n_out % 4 == 0, n_out <= 256.
// As a tiled MFMA GEMM this shape is all fixed cost (one K stage per tile: 20.8 us at 24.5 k x 256, K = 26) and the LayerNorm
// is a second pass over z (11.6 us).  Here a workgroup owns 64 consecutive rows: W^T ([Kp][n_out]) and the rows' inputs
// ([64][Kp], Kp = K rounded up to 4, padding zero) are staged in LDS once -- every global latency of the workgroup is paid in
// that prologue; lane l owns output columns 4 l .. 4 l + 3 of four rows at a time, a k step of four is 4 + 4 ds_read_b128
// (the input reads are same-address broadcasts) and 32 packed FMAs; LayerNorm statistics, affine and ReLU follow in
// registers with the arithmetic of ln_relu_fwd_vec_kernel.  HBM-bound in principle (N (K + 2 n_out) 4 bytes = 50 MB at 24.5 k x
// 256); measured 24 us (GEMM + LayerNorm launches: 32 us) -- the per-workgroup transposing fill of W^T (each of 383
// workgroups walks all 256 rows of W) and the short dependent phases of a 16-rows-per-wave workgroup are what is left.
constexpr int SMALLK_MAX = 64;
constexpr int SMALLK_ROWS = 4;            // rows per wave step
constexpr int SMALLK_BLOCK_ROWS = 64;     // rows per workgroup: 4 waves x 4 steps x 4 rows

__global__ void __launch_bounds__(256)
sage_smallk_fwd_kernel(const float* __restrict__ a1, int64_t lda1, int k1, const float* __restrict__ a2, int64_t lda2, int k2,
                       const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu,
                       float* __restrict__ z_save, int64_t ldz, float* __restrict__ stats, float* __restrict__ y, int64_t ldy,
                       int M, int n, char* __restrict__ yp3 = nullptr, int64_t ldyp3 = 0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = k1 + k2, Kp = (K + 3) & ~3;
    const int ns = n;                                                  // W^T row stride
    float* Wt = sm;                                                    // [Kp][ns], rows K .. Kp-1 zero
    float* xs = sm + Kp * ns;                                          // [64][Kp], columns K .. Kp-1 zero
    const int brow0 = blockIdx.x * SMALLK_BLOCK_ROWS;
    // Transposing fill without index divisions: thread c walks row c of W (its own 4 K-byte run: L1 hits after the first
    // touch) and writes column c of W^T -- consecutive lanes, consecutive LDS words.  (Lanes along a W row instead: every LDS
    // write of a wave hits one bank, 4 M conflict cycles per launch; a flat index with / and %: ~50 VALU per element.)
    // All loads of a thread are requested before its first LDS store (unrolled to the maximal K with uniform guards): a
    // load -> store loop with a runtime trip count paid one global latency per element (26 x ~0.5 us: most of the kernel).
    {
        float wv[SMALLK_MAX];
        const float* wr = W + (int64_t)min((int)threadIdx.x, n - 1) * ldw;
#pragma unroll
        for (int k = 0; k < SMALLK_MAX; ++k) wv[k] = k < K ? wr[k] : 0.f;
        float xv[SMALLK_MAX / 4];
        const int rl = threadIdx.x >> 2, part = threadIdx.x & 3;       // 4 threads per input row
        const int r = min(brow0 + rl, M - 1);
        const float* r1 = a1 + (int64_t)r * lda1;
        const float* r2 = a2 ? a2 + (int64_t)r * lda2 : a1;
#pragma unroll
        for (int i = 0; i < SMALLK_MAX / 4; ++i) {
            const int k = part + 4 * i;
            xv[i] = 0.f;
            if (k < k1) xv[i] = r1[k];
            else if (k < K) xv[i] = r2[k - k1];
        }
        if ((int)threadIdx.x < n) {
#pragma unroll
            for (int k = 0; k < SMALLK_MAX; ++k) if (k < Kp) Wt[k * ns + threadIdx.x] = wv[k];
        }
#pragma unroll
        for (int i = 0; i < SMALLK_MAX / 4; ++i) {
            const int k = part + 4 * i;
            if (k < Kp) xs[rl * Kp + k] = xv[i];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = 4 * lane;
    const bool ok = j < n;
    const bool ln = gamma != nullptr;
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f}, be4[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (bias) b4[e] = bias[j + e];
            if (ln) { g4[e] = gamma[j + e]; be4[e] = beta[j + e]; }
        }
    }
    const float inv_n = 1.0f / (float)n;
    const float* wl = Wt + (ok ? j : 0);
    for (int step = 0; step < SMALLK_BLOCK_ROWS / (4 * SMALLK_ROWS); ++step) {
        const int rl0 = (step * 4 + wave) * SMALLK_ROWS;              // first of this wave's four rows, workgroup-local
        const int row0 = brow0 + rl0;
        if (row0 >= M) break;                                          // wave-uniform
        float acc[SMALLK_ROWS][4];
#pragma unroll
        for (int u = 0; u < SMALLK_ROWS; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][e] = b4[e];
        const float* xq = xs + rl0 * Kp;
        for (int kc = 0; kc < Kp; kc += 4) {
            float4 w4[4], x4[SMALLK_ROWS];
#pragma unroll
            for (int u = 0; u < SMALLK_ROWS; ++u) x4[u] = *reinterpret_cast<const float4*>(xq + u * Kp + kc);    // broadcast
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = *reinterpret_cast<const float4*>(wl + (kc + i) * ns);
#pragma unroll
            for (int u = 0; u < SMALLK_ROWS; ++u) {
                const float xk[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[u][0] = fmaf(xk[i], w4[i].x, acc[u][0]); acc[u][1] = fmaf(xk[i], w4[i].y, acc[u][1]);
                    acc[u][2] = fmaf(xk[i], w4[i].z, acc[u][2]); acc[u][3] = fmaf(xk[i], w4[i].w, acc[u][3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SMALLK_ROWS; ++u) {
            const int r = row0 + u;
            if (r >= M) continue;                                      // wave-uniform
            if (!ln) {
                if (ok) {
                    f4u t;
                    t.x = relu ? fmaxf(acc[u][0], 0.f) : acc[u][0]; t.y = relu ? fmaxf(acc[u][1], 0.f) : acc[u][1];
                    t.z = relu ? fmaxf(acc[u][2], 0.f) : acc[u][2]; t.w = relu ? fmaxf(acc[u][3], 0.f) : acc[u][3];
                    *reinterpret_cast<f4u*>(y + (int64_t)r * ldy + j) = t;
                }
                continue;
            }
            if (z_save && ok) {
                f4u t; t.x = acc[u][0]; t.y = acc[u][1]; t.z = acc[u][2]; t.w = acc[u][3];
                *reinterpret_cast<f4u*>(z_save + (int64_t)r * ldz + j) = t;
            }
            float sm_ = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) sm_ += ok ? acc[u][e] : 0.f;
            const float mean = wave_sum(sm_) * inv_n;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = ok ? acc[u][e] - mean : 0.f; q = fmaf(d, d, q); }
            const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
            if (stats && lane == 0) { stats[r] = mean; stats[M + r] = rstd; }
            if (ok) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = ln_affine((acc[u][e] - mean) * rstd, g4[e], be4[e]);
                    if (relu) o[e] = fmaxf(o[e], 0.f);
                }
                f4u t; t.x = o[0]; t.y = o[1]; t.z = o[2]; t.w = o[3];
                if (y) *reinterpret_cast<f4u*>(y + (int64_t)r * ldy + j) = t;
                // the next (planes) layer's input image, written here instead of by a conversion pass over y
                if (yp3) p3::store4(yp3 + (int64_t)r * ldyp3, j, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

bool smallk_supported(int64_t K, int64_t n_out) {
    static const bool off = GTE_MEASURE_OFF("GTE_SMALLK");
    return !off && K >= 1 && K <= SMALLK_MAX && n_out % 4 == 0 && n_out >= 4 && n_out <= 256;
}

void launch_ln_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu, float* y,
                   int64_t ldy, float* stats, int64_t M, int64_t n_out, hipStream_t s) {
    if (n_out % 4 == 0 && n_out >= 128 && n_out <= 512) {          // 16-byte accesses, two rows per wave
        const dim3 grid((unsigned)gte::ceil_div(M, 8)), block(256);
        if (n_out <= 256)
            hipLaunchKernelGGL((ln_relu_fwd_vec_kernel<1>), grid, block, 0, s, z, ldz, gamma, beta, eps, relu, y, ldy, stats, (int)M,
                               (int)n_out);
        else
            hipLaunchKernelGGL((ln_relu_fwd_vec_kernel<2>), grid, block, 0, s, z, ldz, gamma, beta, eps, relu, y, ldy, stats, (int)M,
                               (int)n_out);
        return;
    }
    hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3((unsigned)gte::ceil_div(M, 4)), dim3(256), 0, s, z, ldz, gamma, beta, eps, relu, y,
                       ldy, stats, (int)M, (int)n_out);
}

}  // namespace

// ------------------------------------------ C ABI -------------------------------------------------
extern "C" int gte_gemm_set_mode(int mode) {
    if (mode != GTE_GEMM_F32 && mode != GTE_GEMM_SPLIT_BF16) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_set_mode: unknown mode %d", mode);
    g_gemm_mode.store(mode, std::memory_order_relaxed);
    return GTE_OK;
}

extern "C" int gte_gemm_get_mode(void) { return gemm_mode(); }

extern "C" int gte_gemm_set_thread_mode(int mode) {
    if (mode != -1 && mode != GTE_GEMM_F32 && mode != GTE_GEMM_SPLIT_BF16)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_set_thread_mode: unknown mode %d", mode);
    t_gemm_mode = mode;
    return GTE_OK;
}

extern "C" int64_t gte_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t mfma = gemm_workspace(M, N, K, 0);
    const int64_t skinny = (M > 0 && M <= SK_MAX) ? gte::round_up(gte::ceil_div(K > 0 ? K : 1, SK_ROWS) * M * N * 4, 256) : 0;
    return mfma > skinny ? mfma : skinny;
}

static int gemm_f32_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace,
                         int64_t workspace_bytes, void* stream, bool may_defer) {
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
    return run_gemm(!trans_a, trans_b != 0, p, workspace, workspace_bytes, s, may_defer);
}

extern "C" int gte_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    return gemm_f32_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, accumulate, workspace, workspace_bytes, stream,
                         false);
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
        return run_gemm(false, false, p, workspace, workspace_bytes, gte::as_stream(stream), true);
    }
    // two launches share ONE workspace: the first fold must have run before the second launch overwrites the slabs
    int rc = gemm_f32_impl(1, 0, n_out, k1, n_nodes, dz, lddz, x1, ldx1, dW, lddw, 0, workspace, workspace_bytes, stream,
                           k2 == 0);
    if (rc != GTE_OK || k2 == 0) return rc;
    return gemm_f32_impl(1, 0, n_out, k2, n_nodes, dz, lddz, x2, ldx2, dW + k1, lddw, 0, workspace, workspace_bytes, stream,
                         true);
}

// ---- transform-then-aggregate ("q-form") entry points ---------------------------------------------------------
// The reference layer computes z = [h | norm * A_w h] W^T + b (models.py:53-72).  With W = [W_s | W_n] and the
// aggregation linear, z = h W_s^T + b + norm * A_w (h W_n^T): when the layer narrows (831 -> 256) aggregating AFTER
// the transform moves 256 columns instead of 831.  In the backward, with q = A_w^T (norm * dz):
//     dW = [dz^T h | q^T h],   dh = dz W_s + q W_n
// so the aggregated input never has to be kept for the backward and dh is ONE GEMM with K = 2 * n_out.
extern "C" int gte_sage_transform_fwd(const float* x, int64_t ldx, int64_t n_feat, const float* W, int64_t ldw,
                                      const float* bias, int64_t n_out, float* t, int64_t ldt, int64_t n_nodes,
                                      void* stream) {
    if (n_nodes < 0 || n_out <= 0 || n_feat <= 0 || n_nodes > INT32_MAX || 2 * n_out > INT32_MAX || n_feat > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_transform_fwd: bad sizes");
    if (n_nodes == 0) return GTE_OK;
    if (!x || !W || !t) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_transform_fwd: null pointer");
    if (ldx < n_feat || ldw < 2 * n_feat || ldt < 2 * n_out)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_transform_fwd: leading dimension too small");
    // t[:, 0:n_out] = x W_s^T + b,  t[:, n_out:2 n_out] = x W_n^T : one launch, two N segments of B = W (stored [n_out][2F])
    GemmParams p = {};
    p.A1 = x; p.lda1 = ldx; p.K1 = (int)n_feat;
    p.B = W; p.ldb = ldw; p.Bn2 = W + n_feat; p.ldbn2 = ldw; p.Nseg = (int)n_out;
    p.C = t; p.ldc = ldt; p.bias = bias; p.bias_cols = (int)n_out; p.M = (int)n_nodes; p.N = (int)(2 * n_out);
    hipStream_t s = gte::as_stream(stream);
    if (needs_small_path(true, true, p)) return run_small(true, true, p, s);
    if (p.lda1 >= GEMM_MAX_LD || p.ldb >= GEMM_MAX_LD)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_transform_fwd: leading dimensions must be < 2^22 elements");
    Plan pl = make_plan(n_nodes, 2 * n_out, n_feat, 0, n_out);
    pl.splits = 1;                                          // K = n_feat is short and M is the node count: never split
    pl.tiles_per_split = (int)gte::ceil_div(n_feat, BK);
    p.splits = 1; p.tiles_per_split = pl.tiles_per_split; p.slab = nullptr;
    return launch_shape<true, true>(p, pl, s);
}

extern "C" int64_t gte_sage_qform_dw_workspace_bytes(int64_t n_out, int64_t n_feat, int64_t n_nodes) {
    return gemm_workspace(n_out, 2 * n_feat, n_nodes, 0, n_feat);
}

// dW[n_out, 2F] = [dz^T x | q^T x]   (dz, q: [nodes][n_out]; x: [nodes][F])
extern "C" int gte_sage_qform_dw(const float* dz, int64_t lddz, const float* q, int64_t ldq, const float* x, int64_t ldx,
                                 int64_t n_feat, float* dW, int64_t lddw, int64_t n_out, int64_t n_nodes, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (n_out <= 0 || n_feat <= 0 || n_nodes < 0 || n_out > INT32_MAX || 2 * n_feat > INT32_MAX || n_nodes > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dw: bad sizes");
    if (!dz || !q || !x || !dW) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dw: null pointer");
    if (lddz < n_out || ldq < n_out || ldx < n_feat || lddw < 2 * n_feat)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dw: leading dimension too small");
    if (n_nodes == 0) {
        for (int64_t r = 0; r < n_out; ++r)
            if (hipMemsetAsync(dW + r * lddw, 0, (size_t)(2 * n_feat) * 4, gte::as_stream(stream)) != hipSuccess)
                return gte::fail(GTE_ERR_LAUNCH, "sage_qform_dw: memset failed");
        return GTE_OK;
    }
    GemmParams p = {};
    p.A1 = dz; p.lda1 = lddz; p.K1 = (int)n_nodes; p.An2 = q; p.ldan2 = ldq;
    p.B = x; p.ldb = ldx; p.Bn2 = x; p.ldbn2 = ldx; p.Nseg = (int)n_feat;
    p.C = dW; p.ldc = lddw; p.M = (int)n_out; p.N = (int)(2 * n_feat);
    return run_gemm(false, false, p, workspace, workspace_bytes, gte::as_stream(stream), true);
}

// dx[nodes, F] = dz W_s + q W_n   (W stored [n_out][2F]: W_s = W[:, 0:F], W_n = W[:, F:2F])
extern "C" int gte_sage_qform_dx(const float* dz, int64_t lddz, const float* q, int64_t ldq, const float* W, int64_t ldw,
                                 int64_t n_feat, int64_t n_out, float* dx, int64_t lddx, int64_t n_nodes, void* stream) {
    if (n_out <= 0 || n_feat <= 0 || n_nodes < 0 || 2 * n_out > INT32_MAX || n_feat > INT32_MAX || n_nodes > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dx: bad sizes");
    if (n_nodes == 0) return GTE_OK;
    if (!dz || !q || !W || !dx) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dx: null pointer");
    if (lddz < n_out || ldq < n_out || ldw < 2 * n_feat || lddx < n_feat)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_qform_dx: leading dimension too small");
    GemmParams p = {};
    p.A1 = dz; p.lda1 = lddz; p.K1 = (int)n_out; p.A2 = q; p.lda2 = ldq; p.K2 = (int)n_out;
    p.B = W; p.ldb = ldw; p.B2 = W + n_feat;               // B(k, n) = W[k][n] for the dz rows, W[k][F + n] for the q rows
    p.C = dx; p.ldc = lddx; p.M = (int)n_nodes; p.N = (int)n_feat;
    hipStream_t s = gte::as_stream(stream);
    if (needs_small_path(true, false, p)) return run_small(true, false, p, s);
    if (p.lda1 >= GEMM_MAX_LD || p.lda2 >= GEMM_MAX_LD || p.ldb >= GEMM_MAX_LD)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_qform_dx: leading dimensions must be < 2^22 elements");
    Plan pl = make_plan(n_nodes, n_feat, n_out, n_out);
    pl.splits = 1;
    pl.tiles_per_split = (int)(2 * gte::ceil_div(n_out, BK));
    p.splits = 1; p.tiles_per_split = pl.tiles_per_split; p.slab = nullptr;
    return launch_shape<true, false>(p, pl, s);
}

static int sage_linear_fwd_impl(const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2,
                                int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                const float* beta, float eps, int relu, float* z_save, int64_t ldz, float* stats,
                                float* y, int64_t ldy, int64_t M, int64_t n_out, void* stream, char* yp3, int64_t ldyp3) {
    if (M < 0 || n_out <= 0 || k1 <= 0 || k2 < 0 || M > INT32_MAX || n_out > INT32_MAX || k1 + k2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: bad sizes");
    if (M == 0) return GTE_OK;
    if (!a1 || !W || (!y && !yp3) || (k2 > 0 && !a2)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: null pointer");
    if (lda1 < k1 || (k2 > 0 && lda2 < k2) || ldw < k1 + k2 || (y && ldy < n_out) || (z_save && ldz < n_out))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: leading dimension too small");
    if (yp3 && !(gamma && smallk_supported(k1 + k2, n_out) && n_out % 16 == 0 && ldyp3 >= p3::row_bytes(n_out) && ldyp3 % 16 == 0))
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_linear_fwd_p3: the image output needs the one-pass form (k1 + k2 <= 64, LayerNorm) "
                                              "and n_out %% 16 == 0");
    if (gamma && !beta) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: gamma without beta");
    hipStream_t s = gte::as_stream(stream);
    const bool ln = gamma != nullptr;
    if (smallk_supported(k1 + k2, n_out) && !(!ln && z_save && z_save != y)) {
        const int64_t kp = (k1 + k2 + 3) & ~(int64_t)3;
        hipLaunchKernelGGL(sage_smallk_fwd_kernel, dim3((unsigned)gte::ceil_div(M, SMALLK_BLOCK_ROWS)), dim3(256),
                           (size_t)(kp * n_out + SMALLK_BLOCK_ROWS * kp) * sizeof(float), s, a1, lda1, (int)k1, a2, lda2, (int)k2,
                           W, ldw, bias, gamma, beta, eps, relu, z_save, ldz, stats, y, ldy, (int)M, (int)n_out, yp3, ldyp3);
        return gte::check_launch("sage_smallk_fwd");
    }
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
        launch_ln_fwd(zbuf, ldzz, gamma, beta, eps, relu, y, ldy, stats, M, n_out, s);
        return gte::check_launch("ln_relu_fwd");
    }
    return GTE_OK;
}

extern "C" int gte_sage_linear_fwd_fuses_ln(int64_t k_total, int64_t n_out) { return smallk_supported(k_total, n_out) ? 1 : 0; }

extern "C" int gte_ln_relu_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu,
                               float* y, int64_t ldy, float* stats, int64_t M, int64_t n_out, void* stream) {
    if (M < 0 || n_out <= 0 || M > INT32_MAX || n_out > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: bad sizes");
    if (M == 0) return GTE_OK;
    if (!z || !y || !gamma || !beta) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: null pointer");
    if (ldz < n_out || ldy < n_out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd: ld < n_out");
    launch_ln_fwd(z, ldz, gamma, beta, eps, relu, y, ldy, stats, M, n_out, gte::as_stream(stream));
    return gte::check_launch("ln_relu_fwd");
}

// ... any width up to 1024 on PADDED rows (ldz, ldy >= n_out rounded up to 4), y as fp32 (nullable) and / or as a P3 image
// (nullable): the LayerNorm(+ReLU) of an aggregate-first planes layer, whose output feeds the next layer's planes GEMM
extern "C" int gte_ln_relu_fwd_p3(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu,
                                  float* y, int64_t ldy, void* yp3, int64_t ldyp3, float* stats, int64_t M, int64_t n_out,
                                  void* stream) {
    if (M < 0 || n_out <= 0 || M > INT32_MAX || n_out > 1024)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd_p3: bad sizes (n_out <= 1024)");
    if (M == 0) return GTE_OK;
    if (!z || !gamma || !beta || (!y && !yp3)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd_p3: null pointer");
    const int64_t n4 = gte::round_up(n_out, 4);
    if (ldz < n4 || (y && ldy < n4) || (yp3 && (ldyp3 < p3::row_bytes(n_out) || ldyp3 % 16 != 0)))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_fwd_p3: rows must be padded to a multiple of 4 floats; ldp >= 96 ceil(n_out / 16)");
    hipStream_t s = gte::as_stream(stream);
    char* yp = reinterpret_cast<char*>(yp3);
    const int64_t n16 = gte::round_up(n_out, 16);
#define GTE_LNF(NV, RF)                                                                                                      \
    hipLaunchKernelGGL((ln_relu_fwd_gen_kernel<NV, RF>), dim3((unsigned)gte::ceil_div(M, 4 * RF)), dim3(256), 0, s, z, ldz, gamma, beta, \
                       eps, relu, y, ldy, yp, ldyp3, stats, (int)M, (int)n_out)
    if (n16 <= 256) GTE_LNF(1, 2); else if (n16 <= 512) GTE_LNF(2, 2); else if (n16 <= 768) GTE_LNF(3, 1); else GTE_LNF(4, 1);
#undef GTE_LNF
    return gte::check_launch("ln_relu_fwd_p3");
}

extern "C" int64_t gte_ln_relu_bwd_workspace_bytes(int64_t M, int64_t n_out) {
    return gte::round_up((int64_t)ln_bwd_blocks(M > 0 ? M : 1) * 3 * (n_out > 0 ? n_out : 1) * 4, 256);
}

static int ln_relu_bwd_impl(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats,
                            const float* gamma, const float* beta, int relu, float* dz, int64_t lddz,
                            float* dgamma, float* dbeta, float* dbias, int64_t M, int64_t n_out, void* workspace,
                            int64_t workspace_bytes, void* stream, char* dzp3, int64_t ldp3) {
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
#define GTE_LNV(NV)                                                                                                 \
    hipLaunchKernelGGL((ln_relu_bwd_vec_kernel<NV>), grid, block, (size_t)(4 * 3 * NV * 256) * sizeof(float), s, dy, lddy, \
                       zz, ldzz, stats, gamma, beta, relu, dz, lddz, part, (int)M, (int)n_out, dzp3, ldp3)
    if (dzp3 && (ldp3 < p3::row_bytes(n_out) || ldp3 % 16 != 0))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd: the P3 image needs ldp >= 96 ceil(n_out / 16)");
    if (dzp3 && !(n_out % 16 == 0 && n_out >= 128 && n_out <= 512)) {
        // any other width up to 1024 with the image: padded rows (ld >= n_out rounded up to 4), LayerNorm layers only
        const int64_t n4 = gte::round_up(n_out, 4);
        if (!gamma || n_out > 1024 || lddy < n4 || ldzz < n4 || lddz < n4)
            return gte::fail(GTE_ERR_UNSUPPORTED, "ln_relu_bwd: the P3 image at this width needs LayerNorm, n_out <= 1024 and rows padded "
                                                  "to a multiple of 4 floats");
#define GTE_LNG(NV, RF)                                                                                                  \
    hipLaunchKernelGGL((ln_relu_bwd_gen_kernel<NV, RF>), grid, block, (size_t)(4 * 3 * NV * 256) * sizeof(float), s, dy, lddy, \
                       zz, ldzz, stats, gamma, beta, relu, dz, lddz, part, (int)M, (int)n_out, dzp3, ldp3)
        const int64_t n16 = gte::round_up(n_out, 16);
        if (n16 <= 256) GTE_LNG(1, 2); else if (n16 <= 512) GTE_LNG(2, 2); else if (n16 <= 768) GTE_LNG(3, 1); else GTE_LNG(4, 1);
#undef GTE_LNG
    } else
    if (n_out % 4 == 0 && n_out >= 128 && n_out <= 512) {              // 16-byte accesses
        if (n_out <= 256) GTE_LNV(1); else GTE_LNV(2);
    } else
    if (n_out <= 64) GTE_LNB(1);
    else if (n_out <= 128) GTE_LNB(2);
    else if (n_out <= 256) GTE_LNB(4);
    else if (n_out <= 512) GTE_LNB(8);
    else if (n_out <= 1024) GTE_LNB(16);
    else
        hipLaunchKernelGGL(ln_relu_bwd_wide_kernel, grid, block, 0, s, dy, lddy, zz, ldzz, stats, gamma, beta, relu, dz,
                           lddz, part, (int)M, (int)n_out);
#undef GTE_LNB
#undef GTE_LNV
    if (dgamma || dbeta || dbias) {
        // deferred (gte_fold_defer_begin): the three column sums join the step's fold batch
        if (gte::defer_fold(part, 3 * n_out, nb, 1, (int)n_out, dgamma, n_out)) {
            gte::defer_fold(part + n_out, 3 * n_out, nb, 1, (int)n_out, dbeta, n_out);
            gte::defer_fold(part + 2 * n_out, 3 * n_out, nb, 1, (int)n_out, dbias, n_out);
        } else {
            hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)gte::ceil_div(n_out, 64)), dim3(1024), 0, s, part, nb,
                               (int)n_out, dgamma, dbeta, dbias);
        }
    }
    return gte::check_launch("ln_relu_bwd");
}

extern "C" int gte_ln_relu_bwd(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats,
                               const float* gamma, const float* beta, int relu, float* dz, int64_t lddz,
                               float* dgamma, float* dbeta, float* dbias, int64_t M, int64_t n_out, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return ln_relu_bwd_impl(dy, lddy, z, ldz, stats, gamma, beta, relu, dz, lddz, dgamma, dbeta, dbias, M, n_out, workspace,
                            workspace_bytes, stream, nullptr, 0);
}

// ... dz additionally as a P3 image (csrc/p3.h): the A operand of the layer's dX / dW planes GEMMs
extern "C" int gte_ln_relu_bwd_p3(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats,
                                  const float* gamma, const float* beta, int relu, float* dz, int64_t lddz, void* dzp3,
                                  int64_t ldp3, float* dgamma, float* dbeta, float* dbias, int64_t M, int64_t n_out,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dzp3) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "ln_relu_bwd_p3: null image");
    return ln_relu_bwd_impl(dy, lddy, z, ldz, stats, gamma, beta, relu, dz, lddz, dgamma, dbeta, dbias, M, n_out, workspace,
                            workspace_bytes, stream, reinterpret_cast<char*>(dzp3), ldp3);
}

extern "C" int gte_sage_linear_fwd(const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2,
                                   int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                   const float* beta, float eps, int relu, float* z_save, int64_t ldz, float* stats,
                                   float* y, int64_t ldy, int64_t M, int64_t n_out, void* stream) {
    if (!y) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd: null pointer");
    return sage_linear_fwd_impl(a1, lda1, k1, a2, lda2, k2, W, ldw, bias, gamma, beta, eps, relu, z_save, ldz, stats, y, ldy, M, n_out,
                                stream, nullptr, 0);
}

// ... the one-pass form (gte_sage_linear_fwd_fuses_ln) with y written as a P3 image for the next layer's planes GEMM (y itself
// may then be NULL)
extern "C" int gte_sage_linear_fwd_p3(const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2,
                                      int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                      const float* beta, float eps, int relu, float* z_save, int64_t ldz, float* stats,
                                      float* y, int64_t ldy, void* yp3, int64_t ldyp3, int64_t M, int64_t n_out, void* stream) {
    if (!yp3) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_linear_fwd_p3: null image");
    return sage_linear_fwd_impl(a1, lda1, k1, a2, lda2, k2, W, ldw, bias, gamma, beta, eps, relu, z_save, ldz, stats, y, ldy, M, n_out,
                                stream, reinterpret_cast<char*>(yp3), ldyp3);
}
