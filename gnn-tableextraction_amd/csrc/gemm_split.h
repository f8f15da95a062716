// fp32 GEMM on the bf16 matrix pipe: the "split" mode of the transform GEMMs (gte_gemm_set_mode(GTE_GEMM_SPLIT_BF16)).
// Included by sage_linear.hip inside its anonymous namespace (GemmParams, static_for, SRD_FLAGS, f32x4, f32x16 are in scope).
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 retires 64 flop / cycle / SIMD, v_mfma_f32_32x32x16_bf16 1024: the fp32 matrix rate
// is 1/16 of the bf16 rate.  Every fp32 operand x is cut EXACTLY into three bf16 pieces when its tile goes registers -> LDS,
//     h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round to nearest even; the subtractions are exact)
//     |m| <= 2^-8 |x|,  |l| <= 2^-16 |x|,  |x - (h + m + l)| <= 2^-24 |x|  (3 x 8 significand bits),
// and a product a b is formed as  hh + hm + mh + mm + hl + lh  (six bf16 MFMAs; each piece product is exact in fp32, the sum
// runs in the fp32 accumulator, smallest terms first).  Dropped: ml + lm + ll <= 2^-23 |a b| worst case, ~2^-25 typical --
// below the rounding an fp32 FMA chain commits on every step.  Measured against fp64 on random and on training operands
// (profiles/r02/gemm_split.md, tests/test_gemm_split.py): error <= the native fp32 MFMA kernel's on every shape.
// Six 32-cycle MFMAs per K = 16 against eight 64-cycle MFMAs: 2.67x the matrix-pipe rate at the same accuracy class.
// Non-finite operands: inf - inf in the split makes NaN where the fp32 kernel would return inf.  Operands below ~2^-109 in
// magnitude: the low piece (2^-17 of the operand) leaves bf16's normal range; tests cover operand scales 2^-60 .. 2^60.
//
// Structure (one workgroup = 4 waves as 2 x 2, block tile BM x 128, 2 workgroups per CU):
//   stage = 16 k.  global -> registers: buffer loads through the same range-checked windows as the fp32 kernel (zero fill
//   past the matrix edge / the K range), two register sets, tile t+2 requested at the START of stage t;
//   registers -> LDS: split + 8-byte stores into three piece planes per operand, double buffered, one barrier per stage;
//   K-contiguous operands: planes [rows][16 + 8 pad] bf16, fragment = one ds_read_b128 (a lane's 8 consecutive k);
//   row-contiguous operands ([K][rows] in memory): planes [16][rows + 32 pad] -- the memory order, coalesced stores -- and the
//   fragment is two ds_read_b64_tr_b16 (the LDS transposes: lane i of a 16-lane group receives column i of a 4 x 16 block);
//   the barrier sits two products before the end of the stage: the next stage's fragment reads run under the last 8 MFMAs.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KS = 16;                                     // k per stage (one 32x32x16 MFMA step)
#ifndef SPLIT_PIN
#define SPLIT_PIN 1                                        // 0: compiler's own schedule, 2: nothing moves across the barrier
#endif
#ifndef SPLIT_NQ1
#define SPLIT_NQ1 4                                        // products of a stage issued before its barrier (of 6)
#endif

// two fp32 -> their packed bf16 pieces.  The residuals are separate scalar subtractions on purpose: a packed fp32 add
// (v_pk_add_f32, what a 2-vector subtraction compiles to) issued beside MFMAs costs ~13 cycles more than two plain ones.
__device__ __forceinline__ float plain_sub(float a, float b) {
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_pair(const float x0, const float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{x0, x1}), bf16x2));
    const float r10 = plain_sub(x0, __builtin_bit_cast(float, h << 16));
    const float r11 = plain_sub(x1, __builtin_bit_cast(float, h & 0xffff0000u));
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{r10, r11}), bf16x2));
    const float r20 = plain_sub(r10, __builtin_bit_cast(float, m << 16));
    const float r21 = plain_sub(r11, __builtin_bit_cast(float, m & 0xffff0000u));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{r20, r21}), bf16x2));
}

// One operand of the block tile, ROWS rows x KS.  KC: stored [rows][K] in memory, else [K][rows].
//   KC : chunk i = (row r0 + 64 i, k = 4 kq .. 4 kq + 3), kq = tid % 4, r0 = tid / 4       plane [ROWS][48 bytes]
//   !KC: chunk i = (k = kr0 + i KSTEP, rows 4 rq .. 4 rq + 3), rq = tid % (ROWS / 4)      plane [KS][ROWS * 2 + 64 bytes]
// (row stride of the !KC plane = 64 mod 128 bytes: the four k rows a transposed read touches fall on disjoint banks)
template <bool KC, int ROWS>
struct SplitOperand {
    static constexpr int NCH = ROWS * (KS / 4) / 256;
    static constexpr int CPR = ROWS / 4, KSTEP = 256 / CPR;
    static constexpr int RS = ROWS * 2 + 64;
    static constexpr int PLANE = KC ? ROWS * 48 : KS * RS;            // bytes
    static constexpr int LDS_BYTES = 3 * PLANE;
    static constexpr int CHUNK_STRIDE = KC ? 64 * 48 : KSTEP * RS;
    static_assert(NCH >= 1 && (ROWS == 64 || ROWS == 128), "tile rows");
    int voff[2][NCH];        // byte offset of chunk i inside the window, per K segment
    int wofs;                // byte offset of chunk 0 inside a plane
    int kpos;                // KC: k offset of this lane's chunks inside a stage
    int rd;                  // byte offset of this lane's fragment inside a plane (32-row block 0)
    f32x4 r[2][NCH];         // two staging register sets
    int vc[2];               // KC: valid k count of the staged chunks, per set

    __device__ __forceinline__ void init(int64_t ld0, int64_t ld1, int tid, int lane, int blk_row0) {
        if constexpr (KC) {
            // rows of a 16-lane group (the unit a ds_write_b64 is served in) are 2 apart: with 48-byte rows the four 32-byte
            // row segments of rows r, r+2, r+4, r+6 fall on disjoint bank octets (consecutive rows: r and r+3 overlap --
            // SQ_LDS_BANK_CONFLICT 7.8 M cycles per layer-0 forward launch, 0 after)
            const int kq = tid & 3, r0 = 8 * (tid >> 5) + ((tid >> 4) & 1) + 2 * ((tid >> 2) & 3);
            kpos = kq * 4;
            wofs = r0 * 48 + kq * 8;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                voff[0][i] = (int)(((r0 + 64 * i) * ld0 + kq * 4) * 4);
                voff[1][i] = (int)(((r0 + 64 * i) * ld1 + kq * 4) * 4);
            }
            rd = (blk_row0 + (lane & 31)) * 48 + (lane >> 5) * 16;
        } else {
            const int rq = tid % CPR, kr0 = tid / CPR;
            kpos = 0;
            wofs = kr0 * RS + rq * 8;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                voff[0][i] = (int)(((kr0 + i * KSTEP) * ld0 + rq * 4) * 4);
                voff[1][i] = (int)(((kr0 + i * KSTEP) * ld1 + rq * 4) * 4);
            }
            // lane = 32 h + 16 g + 4 q + p supplies the address of k row 8 h + q, rows 16 g + 4 p .. + 3 of the block
            const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
            rd = (8 * h + q) * RS + (blk_row0 + 16 * g + 4 * pp) * 2;
        }
        vc[0] = vc[1] = 4;
    }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t window(const float* origin, int ld, int kl, int kseg, int rows_valid) const {
        const int left = kseg - kl;
        const int pos = left > 0 ? 1 : 0;
        if constexpr (KC) {
            const int bytes = ((rows_valid - 1) * ld + left) * 4 * pos;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin + kl), 0, bytes, SRD_FLAGS);
        } else {
            const int krows = left < KS ? left : KS;
            const int bytes = ((krows - 1) * ld + rows_valid) * 4 * pos;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin + (int64_t)kl * ld), 0, bytes, SRD_FLAGS);
        }
    }
    template <int SET, int i>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t srd, int seg, int oob) {
        const int vo = seg ? voff[1][i] : voff[0][i];
        r[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, KC ? (vo | oob) : vo, 0, 0));
    }
    // chunk i of set SET -> its three pieces in the planes at `planes`
    // MASK: the K ranges are not multiples of 4, so a chunk can straddle the end of its row: keep the valid prefix
    template <int SET, int i, bool MASK>
    __device__ __forceinline__ void store(char* __restrict__ planes) const {
        f32x4 v = r[SET][i];
        if constexpr (KC && MASK) {
            v.y = vc[SET] > 1 ? v.y : 0.f;
            v.z = vc[SET] > 2 ? v.z : 0.f;
            v.w = vc[SET] > 3 ? v.w : 0.f;
        }
        uint2 h, m, l;
        split_pair(v.x, v.y, h.x, m.x, l.x);
        split_pair(v.z, v.w, h.y, m.y, l.y);
        char* dst = planes + wofs + i * CHUNK_STRIDE;
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + PLANE) = m;
        *reinterpret_cast<uint2*>(dst + 2 * PLANE) = l;
    }
    // fragment of piece `piece`, 32-row block `blk` (relative to the block rd was built for)
    __device__ __forceinline__ bf16x8 frag(const char* __restrict__ planes, int piece, int blk) const {
        if constexpr (KC) {
            return *reinterpret_cast<const bf16x8*>(planes + piece * PLANE + rd + blk * 32 * 48);
        } else {
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            const char* a = planes + piece * PLANE + rd + blk * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 4 * RS));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    }
};

template <bool AK, bool BKC, int BM, int BN>
constexpr int gemm_split_lds_bytes() { return 2 * (SplitOperand<AK, BM>::LDS_BYTES + SplitOperand<BKC, BN>::LDS_BYTES); }

template <int TM, int TN>
struct SplitFrags { bf16x8 a[3][TM], b[3][TN]; };

// Same contract as gemm_f32_mfma_kernel (tile mapping, K segments, split-K slabs, tail pieces, epilogue): the two are
// interchangeable per launch.
// KTAIL: some K segment length is not a multiple of 4 (only then K-contiguous chunks need their tail masks: 3 selects per
// chunk and stage; the 256- and 512-deep GEMMs of the step run without them)
template <bool AK, bool BKC, int BM, int BN, bool KTAIL>
__global__ void __launch_bounds__(256, 2)
gemm_split_kernel(const GemmParams p) {
    constexpr int WN = 2;
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert(TM >= 1 && TN >= 1, "2 x 2 waves");
    using OpA = SplitOperand<AK, BM>;
    using OpB = SplitOperand<BKC, BN>;
    constexpr int BUF = OpA::LDS_BYTES + OpB::LDS_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- work unit -> (logical tile, K range): as gemm_f32_mfma_kernel ----
    const int seg_cols = p.Bn2 ? p.Nseg : p.N;
    const int seg_tiles = (seg_cols + BN - 1) / BN;
    const int tiles_n = p.Bn2 ? 2 * seg_tiles : seg_tiles, tiles_m = (p.M + BM - 1) / BM;
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    const bool tail = p.sk_pieces > 1 && (int)blockIdx.x >= p.sk_full;
    const int tail_j = tail ? ((int)blockIdx.x - p.sk_full) / p.sk_pieces : 0;
    const int tail_p = tail ? ((int)blockIdx.x - p.sk_full) % p.sk_pieces : 0;
    const unsigned nremap = p.sk_pieces > 1 ? (unsigned)p.sk_full : ntile * (unsigned)(p.splits > 1 ? p.splits : 1);
    const unsigned lu = tail ? (unsigned)(p.sk_full + tail_j) : gte_xcd_remap(blockIdx.x, nremap);
    const unsigned lb = lu % ntile;
    const int split = (int)(lu / ntile);
    const int tm = lb / tiles_n, tn = lb % tiles_n;
    const int nseg = tn / seg_tiles;
    const int m0 = tm * BM, n0 = (tn % seg_tiles) * BN;
    const float* Bmat = nseg ? p.Bn2 : p.B;
    const int64_t ldbm = nseg ? p.ldbn2 : p.ldb;

    const int tiles_seg0 = (p.K1 + BK - 1) / BK, tiles_seg1 = (p.K2 + BK - 1) / BK;
    const int total_tiles = tiles_seg0 + tiles_seg1;
    const int t_begin = tail ? tail_p * p.sk_tps : split * p.tiles_per_split;
    const int t_end = min(total_tiles, t_begin + (tail ? p.sk_tps : p.tiles_per_split));

    const int rowsA = min(BM, p.M - m0), rowsB = min(BN, seg_cols - n0);
    const float* Amat = (nseg && p.An2) ? p.An2 : p.A1;
    const int64_t ldam = (nseg && p.An2) ? p.ldan2 : p.lda1;
    const float* a_org0 = AK ? Amat + (int64_t)m0 * ldam : Amat + m0;
    const float* a_org1 = p.A2 ? (AK ? p.A2 + (int64_t)m0 * p.lda2 : p.A2 + m0) : a_org0;
    const float* b_org0 = BKC ? Bmat + (int64_t)n0 * ldbm : Bmat + n0;
    const float* b_org1 = p.B2 ? (BKC ? p.B2 + (int64_t)n0 * ldbm : p.B2 + n0)
                               : (BKC ? b_org0 + p.K1 : b_org0 + (int64_t)p.K1 * ldbm);
    const int ld_a0 = (int)ldam, ld_a1 = (int)(p.A2 ? p.lda2 : ldam), ld_b = (int)ldbm;

    OpA opa;
    OpB opb;
    opa.init(ldam, p.A2 ? p.lda2 : ldam, tid, lane, wm * TM * 32);
    opb.init(ldbm, ldbm, tid, lane, wn * TN * 32);
    constexpr int NCA = OpA::NCH, NCB = OpB::NCH, NC = NCA + NCB;

    // ---- K cursor in stages of KS.  A K segment spans whole K tiles of BK (the unit of the split / tail bookkeeping):
    // stages between the end of the segment and the end of its last K tile read an empty window (zeros). ----
    struct StageDesc {
        __amdgpu_buffer_rsrc_t sa, sb;
        int seg, oob_a, oob_b, vca, vcb;
    };
    int c_seg = t_begin >= tiles_seg0 ? 1 : 0;
    int c_kl = (t_begin - (c_seg ? tiles_seg0 : 0)) * BK;
    int c_left = (t_end - t_begin) * (BK / KS);
    int kseg0 = p.K1, kseg1 = p.K2;
    asm volatile("" : "+s"(kseg0), "+s"(kseg1));
    const int span0 = tiles_seg0 * BK, span1 = tiles_seg1 * BK;
    auto describe = [&]() {
        StageDesc d;
        int kseg = c_seg == 0 ? kseg0 : kseg1;
        kseg = (c_seg < 2 && c_left > 0) ? kseg : 0;
        const int span = c_seg == 0 ? span0 : span1;
        const int left = kseg - c_kl;
        d.seg = c_seg == 1 ? 1 : 0;
        d.sa = opa.window(d.seg ? a_org1 : a_org0, d.seg ? ld_a1 : ld_a0, c_kl, kseg, rowsA);
        d.sb = opb.window(d.seg ? b_org1 : b_org0, ld_b, c_kl, kseg, rowsB);
        d.vca = left - opa.kpos;
        d.vcb = left - opb.kpos;
        d.oob_a = ((d.vca - 1) >> 31) & (int)0x80000000;
        d.oob_b = ((d.vcb - 1) >> 31) & (int)0x80000000;
        c_kl += KS;
        --c_left;
        const bool done = c_kl >= span;
        c_seg = done ? (c_seg < 2 ? c_seg + 1 : 2) : c_seg;
        c_kl = done ? 0 : c_kl;
        return d;
    };
    auto issue_loads = [&](auto SET, const StageDesc& d) {
        constexpr int st = decltype(SET)::value;
        static_for<NC>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < NCA) opa.template load<st, j>(d.sa, d.seg, d.oob_a);
            else opb.template load<st, j - NCA>(d.sb, d.seg, d.oob_b);
        });
        opa.vc[st] = d.vca;
        opb.vc[st] = d.vcb;
    };
    auto store_stage = [&](auto SET, char* buf) {
        constexpr int st = decltype(SET)::value;
        static_for<NC>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < NCA) opa.template store<st, j, KTAIL>(buf);
            else opb.template store<st, j - NCA, KTAIL>(buf + OpA::LDS_BYTES);
        });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    using Frags = SplitFrags<TM, TN>;
    auto read_frags = [&](Frags& f, const char* buf) {
#pragma unroll
        for (int pc = 2; pc >= 0; --pc) {                      // l planes first: the first products need them
#pragma unroll
            for (int a = 0; a < TM; ++a) f.a[pc][a] = opa.frag(buf, pc, a);
#pragma unroll
            for (int b = 0; b < TN; ++b) f.b[pc][b] = opb.frag(buf + OpA::LDS_BYTES, pc, b);
        }
    };
    // piece of a / piece of b per product, smallest terms first; the last two (mh, hh) run after the barrier
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
    auto products = [&](const Frags& f, auto Q0, auto Q1) {
#pragma unroll
        for (int q = decltype(Q0)::value; q < decltype(Q1)::value; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IQ = std::integral_constant<int, SPLIT_NQ1>;
    using I6 = std::integral_constant<int, 6>;
    constexpr int NMF1 = SPLIT_NQ1 * TM * TN, NMF2 = (6 - SPLIT_NQ1) * TM * TN;
    constexpr int NVALU = NC * 25 + 16;                        // split (22 / chunk) + tail masks + window arithmetic's vector part
    constexpr int VSLOTS = NMF1 > 4 ? NMF1 - 4 : NMF1;         // the vector work ends 4 MFMAs before the barrier: the LDS stores land
    constexpr int NFR = 3 * (TM * (AK ? 1 : 2) + TN * (BKC ? 1 : 2));   // fragment read instructions per stage
    StageDesc D;                                               // the tile the NEXT stage requests (described one stage ahead)
    // stage: multiply the tile in `cur` (fragments f); the tile in register set R^1 goes to `nxt`; set R takes tile t+2
    auto stage = [&](Frags& f, Frags& fn, char* nxt, auto R) {
        constexpr int rr = decltype(R)::value;
        issue_loads(std::integral_constant<int, rr>{}, D);
        store_stage(std::integral_constant<int, rr ^ 1>{}, nxt);
        products(f, I0{}, IQ{});
        D = describe();
#if SPLIT_PIN
#pragma unroll
        for (int i = 0; i < NMF1; ++i) {
            GTE_SGB(SG_MFMA, 1);
            if (i == 0) GTE_SGB(SG_VMEM_R, NC);
            if (i < VSLOTS) {
                GTE_SGB(SG_VALU, (NVALU + VSLOTS - 1) / VSLOTS);
                GTE_SGB(SG_DS_W, (3 * NC + VSLOTS - 1) / VSLOTS);
            }
            GTE_SGB(SG_SALU, (56 + NMF1 - 1) / NMF1);
        }
#endif
#if SPLIT_PIN == 2
        __builtin_amdgcn_sched_barrier(0);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(fn, nxt);
        products(f, IQ{}, I6{});
#if SPLIT_PIN
#pragma unroll
        for (int i = 0; i < NMF2; ++i) {
            GTE_SGB(SG_MFMA, 1);
            GTE_SGB(SG_DS_R, (NFR + NMF2 - 1) / NMF2);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    if (t_begin < t_end) {
        char* s0 = lds_raw;
        char* s1 = lds_raw + BUF;
        {
            const StageDesc d0 = describe();
            issue_loads(I0{}, d0);
            const StageDesc d1 = describe();
            issue_loads(I1{}, d1);
            store_stage(I0{}, s0);
            D = describe();
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        Frags f0, f1;
        read_frags(f0, s0);
        const int nst = (t_end - t_begin) * (BK / KS);          // even
        for (int t = 0; t < nst; t += 2) {
            stage(f0, f1, s1, I0{});
            stage(f1, f0, s0, I1{});
        }
    }

    // ---- epilogue: as gemm_f32_mfma_kernel (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    if (tail) {
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
#ifdef SPLIT_NOSTORE                                        // measurement only: what the epilogue costs (results are not written)
    if (acc[0][0][0] != 12345.678f) return;
#endif
    float* outp = p.splits > 1 ? p.slab + (int64_t)split * p.M * p.N : p.C;
    const int64_t ldo = p.splits > 1 ? p.N : p.ldc;
    const bool post = p.splits <= 1 && (p.accumulate || p.relu);
    // Branch-free stores through a buffer descriptor over the output: rows past M and columns past the tile's segment get an
    // offset outside the window and are dropped by the range check -- no per-element compare / exec mask / 64-bit address
    // (the guarded pointer version below executes ~11 vector instructions and 2 branches per element: a third of the
    // vector instructions of a 256-deep GEMM).  Needs every offset of the padded tile grid to fit 32 bits.
    const int64_t extent = ((int64_t)p.M + BM) * ldo * 4;
    if (extent < ((int64_t)1 << 31)) {
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
