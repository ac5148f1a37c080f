// Row-resident pointwise input gradient (bf16): the conv1-type input gradients of a bottleneck block -- few contraction
// channels (the block's middle width, 64..512), many output channels (4x that), and an epilogue that streams three
// full-width tensors (the identity branch's gradient, the merge-ReLU mask, the raw output of the previous block's last
// BatchNorm for sum(dz * xhat)).  Reference ops replaced: autograd's backward of torchvision Bottleneck.conv1 + the block
// merge + the previous block's bn3 (main_code/utils/backbones.py:16-18 builds the ResNet-50; model_utils.py:185 runs it).
//
// k_igemm gives such a launch one 128 x 128 tile per block: fill the prologue tables, 2-16 K-chunks, then an epilogue that
// only now asks for its 64 KB of operands -- a serial chain of ~13 us per tile (profiles/r03_igemm_stamps.txt), the BN-backward
// prologue re-evaluated for every column tile (8x at layer3), and 784 tiles on 512 slots.  Here a persistent block owns a
// contiguous range of (64-pixel row block, 128-channel column tile) items, row block major:
//   * the transformed operand dy = alpha * dz + beta * y + gam of a row block is built ONCE and stays in LDS;
//   * the weights of a column tile come straight from L2 into the MFMA fragment registers (a lane's 16 bytes are 8
//     consecutive k of one output channel: no staging), each wave owning 32 of the 128 columns and all 64 pixels;
//   * the epilogue operands of item i + 1 are fetched by LDS-DMA -- every lane fetches exactly the 16 bytes it will
//     consume, so the lane-linear image needs no layout and no barrier -- while item i's epilogue arithmetic and item
//     i + 1's MFMAs run;
//   * the per-channel statistics accumulate in an LDS table (each column has one owning lane) and reach the replicated
//     totals as one burst of atomics per block; sum(dz * xhat) is closed there as invstd * (sum(dz * y) - mean * sum(dz)).
// Ranges are split evenly (items * b / blocks), so every block runs the same count +- 1.
#include "conv_launch.h"

namespace frx {

constexpr int PWR_BM = 64, PWR_BN = 128, PWR_NT = 512;
// one ring slot of epilogue operands: [tensor (addend, raw y)][row fragment][wave] x 1 KiB (lane-linear 16-byte pieces), then
// the merge-ReLU mask bytes as [row fragment][wave] x 256 B (lane-linear dwords: the dword that holds the lane's byte)
constexpr int PWR_EBYTES = 2 * 4 * 4 * 1024 + 4 * 4 * 256;
constexpr int PWR_LUT = 256 * 16;                  // mask byte -> 16-byte AND mask over 8 packed bf16

unsigned pw_rows_lds(int Kc, int Ncol, int ring, int rbn) { return (unsigned)((rbn ? rbn : 1) * PWR_BM * Kc * 2 + ring * PWR_EBYTES + PWR_LUT + 3 * Kc * 4 + 2 * Ncol * 4); }

// (hipcc may park a block-uniform descriptor in vector registers when scalar registers run short; the DMA wants it in SGPRs)
__device__ __forceinline__ u32x4_t pwr_sgpr4(u32x4_t r) {
  u32x4_t o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)r[i]);
  return o;
}
// LDS-DMA of one dword per lane (lane-linear at M0 + 4 * lane), as dma16
__device__ __forceinline__ void dma4(u32x4_t rsrc, unsigned lds_addr, unsigned voff, int soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// The per-item barrier orders LDS traffic only.  __syncthreads() is fence + barrier: its fence drains vmcnt -- the compute
// waves' output stores and, worse, every DMA a loader has in flight (the ring would be one item deep whatever its size).
__device__ __forceinline__ void pwr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XOR swizzle of the 16-byte slot index of a [64 rows][RB bytes] image read with ds_read_b128 in the MFMA fragment pattern
// (row = lane & 15, slot = 4 ks + (lane >> 4)): the 16 lanes of a service group ({fq 0: fr 0-3, 12-15; fq 1: fr 4-11}, ...)
// land on 16 distinct 16-byte positions of the 256-byte bank row.
template <int RB> __device__ __forceinline__ int pwr_swz(int row) {
  if constexpr (RB >= 256) return row & 15;
  else return (row >> 1) & 7;
}

// One block per CU, eight waves: waves 0-3 compute (wave w: all 64 pixels x columns 32 w .. 32 w + 31 of the item's 128),
// waves 4-7 only LOAD -- wave 4 + w keeps RING - 1 items of wave w's epilogue operands in flight by LDS-DMA, lane for lane
// the 16-byte pieces (and the mask dword) lane l of wave w will consume: the lane-linear image needs no layout.  A loader's
// instruction stream holds nothing but those DMAs, so its counted vmcnt is exact; one barrier per item hands a filled slot
// to the compute waves and the slot they just finished back to the loaders.
// The compute waves are the bound (measured: the epilogue arithmetic of an item, ~2 us on one wave per SIMD; MFMA 0.2-0.7),
// so their instruction count is what is tuned: the mask is applied to the PACKED result through a 256-entry LDS table (one
// AND per bf16 pair), and with CTN > 0 -- the launch has CTN <= 4 column tiles -- the per-channel sums of every column tile
// stay in registers for the whole block (one lane-reduction per block instead of per item); with two column tiles the
// weight fragments of both stay in registers as well.
// RBN > 0 (layer2 / layer3: four / eight column tiles, a dozen / half a dozen items per block): the block's range spans at
// most RBN row blocks; ALL of them are built at set-up and stay in LDS, and the items are walked COLUMN TILE MAJOR -- the
// weight fragments are fetched once per column tile instead of once per item (measured: that fetch, queued behind the
// loaders' DMAs, cost 0.8-1.3 us of an item's 3.2) and the lane reduction of the statistics runs once per column tile.
template <int KCH, bool ADD, int RING, int CTN, int RBN>
__global__ __launch_bounds__(PWR_NT, 2) void k_pw_rows_dgrad(ConvArgs a, int items, int col_tiles) {
  constexpr bool COLMAJ = RBN > 0;
  static_assert(!COLMAJ || CTN == 0, "column-tile-major walk: statistics per column tile, weights per column tile");
  typedef bf16_t T;
  constexpr int RB = KCH * 2, SPR = KCH / 8, KS = KCH / 32;
  constexpr int CT = 256;                               // compute threads
  constexpr int ALD = KS, RSTEP = CT / SPR;             // 16-byte loads per compute thread, tensor and row block; rows between them
  constexpr int NE = (ADD ? 8 : 4) + 4;                 // DMA instructions per item and loader wave
  constexpr bool APF = KCH <= 128 && !COLMAJ && !(KCH == 128 && CTN == 2);      // the next row block's raw operand rows are requested an item ahead (registers; not where both column tiles' weights already fill them)
  constexpr bool WRES = CTN == 2;                       // both column tiles' weight fragments resident in registers
  constexpr int NCT = CTN > 0 ? CTN : 1;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(KCH == 64 || KCH == 128 || KCH == 256, "middle widths served");
  static_assert(RING >= 2 && RING <= 5 && (CTN == 0 || CTN == 2), "ring slots; column tiles held in registers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sE = sA + (COLMAJ ? RBN : 1) * PWR_BM * RB;
  char* sLut = sE + RING * PWR_EBYTES;
  float* sTab = reinterpret_cast<float*>(sLut + PWR_LUT);         // [KCH / 8][alpha, beta, gam][8]
  float* sStat = sTab + 3 * KCH;                                  // CTN == 0: [2][Ncol]: sum dz, sum dz * y
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int cw = wave & 3;                              // the compute wave (whose operands a loader fetches)
  const int fr = lane & 15, fq = lane >> 4;
  const int lo = (int)((long)items * blockIdx.x / gridDim.x), hi = (int)((long)items * (blockIdx.x + 1) / gridDim.x);
  if (lo >= hi) return;
  FRX_STAMP(0);
  // COLMAJ: the block's items in column-tile-major order over its row blocks rb_lo .. rb_lo + nrb - 1; (ct, r) is the walk's
  // state, the call returns the next item of the range (or -1).  Loaders and compute waves run the same walk.
  const int rb_lo = lo / col_tiles, nrb = (hi - 1) / col_tiles - rb_lo + 1;
  auto walk = [&](int& ct, int& r) -> int {
    while (ct < col_tiles) {
      while (r < nrb) {
        const int it = (rb_lo + r) * col_tiles + ct;
        ++r;
        if (it >= lo && it < hi) return it;
      }
      r = 0; ++ct;
    }
    return -1;
  };

  const unsigned ybytes = (unsigned)a.M * (unsigned)a.Ncol * 2u;
  const int Hc = (a.Ho + 1) >> 1, Wc = (a.Wo + 1) >> 1, hw = a.Ho * a.Wo;
  const unsigned rstep = 16u * (unsigned)a.Ncol * 2u;   // bytes between the lane's row fragments
  // byte offsets of the lane's four 16-byte pieces of item (m0, n0) in the output-shaped tensors (yo) and in the addend (ao)
  auto row_offsets = [&](int m0, int n0, unsigned (&yo)[4], unsigned (&ao)[4]) {
    const int nb = n0 + 32 * cw + 8 * fq;
    const unsigned y0 = (unsigned)(((m0 + fr) * a.Ncol + nb) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 16 * i + fr;
      yo[i] = m < a.M ? y0 + (unsigned)i * rstep : OOB;
      ao[i] = yo[i];
      if (ADD && a.add_stride == 2) {
        const int n = m / hw, rem = m - n * hw, h = rem / a.Wo, ww = rem - h * a.Wo;
        const bool on = m < a.M && !((h | ww) & 1);
        ao[i] = on ? (unsigned)((((n * Hc + (h >> 1)) * Wc + (ww >> 1)) * a.Ncol + nb) * 2) : OOB;
      }
    }
  };

  if (loader) {
    // ---------------------------------------------------------------- loader waves
    unsigned addbytes = ybytes;
    if (ADD && a.add_stride == 2) addbytes = (unsigned)a.N * Hc * Wc * a.Ncol * 2u;
    const u32x4_t rawEy = pwr_sgpr4(raw_rsrc(a.e_y, ybytes)), rawAdd = pwr_sgpr4(raw_rsrc(ADD ? a.addend : a.e_y, addbytes));
    const u32x4_t rawBits = pwr_sgpr4(raw_rsrc(a.e_bits, ybytes / 16u));
    const unsigned ldsE = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sE + (unsigned)cw * 1024u;
    const unsigned ldsB = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sE + 32768u + (unsigned)cw * 256u;
    auto issue_dma = [&](int it, int seq) {          // seq: the item's position in the block's walk (its ring slot)
      const int rb = it / col_tiles, ct = it - rb * col_tiles;
      const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane(seq % RING) * (unsigned)PWR_EBYTES;
      unsigned yo[4], ao[4];
      row_offsets(rb * PWR_BM, ct * PWR_BN, yo, ao);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ADD) dma16(rawAdd, ldsE + slot + (unsigned)((0 * 4 + i) * 4096), ao[i], 0);
        dma16(rawEy, ldsE + slot + (unsigned)((1 * 4 + i) * 4096), yo[i], 0);
      }
      // mask bytes: byte (m * Ncol + nb) / 8 of the lane's piece sits in the dword the four fq-lanes of its row share
#pragma unroll
      for (int i = 0; i < 4; ++i) dma4(rawBits, ldsB + slot + (unsigned)(i * 1024), yo[i] == OOB ? OOB : ((yo[i] >> 4) & ~3u), 0);
    };
    int lct = 0, lr = 0;              // (COLMAJ) the loaders' position in the walk: RING - 1 items ahead of the compute waves
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) {
      if (lo + k < hi) {
        if constexpr (COLMAJ) issue_dma(walk(lct, lr), k); else issue_dma(lo + k, k);
      }
    }
    __syncthreads();                  // (tables built)
    __syncthreads();                  // (first row block / all row blocks committed)
    int cur_rb = lo / col_tiles;
    const int n = hi - lo;
    for (int k = 0; k < n; ++k) {
      // item k of the walk has landed once at most the DMAs of the items after it are in flight
      if (k + RING - 2 < n) wait_vmcnt<(RING - 2) * NE>(); else wait_vmcnt<0>();
      pwr_barrier();                  // item k handed over; the compute waves are done with item k - 1: its slot is free
      if (k + RING - 1 < n) {
        if constexpr (COLMAJ) issue_dma(walk(lct, lr), k + RING - 1); else issue_dma(lo + k + RING - 1, k + RING - 1);
      }
      if constexpr (!COLMAJ) {
        const int rb = (lo + k) / col_tiles;
        if (rb != cur_rb) { pwr_barrier(); cur_rb = rb; }      // (the compute waves' barrier behind a new row block)
      }
    }
  } else {
    // ---------------------------------------------------------------- compute waves
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X2), 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcDy = __builtin_amdgcn_make_buffer_rsrc(a.dy_out ? a.dy_out : const_cast<void*>(a.X), 0, a.dy_out ? a.xbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, a.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ybytes, 0x00020000);
    // a row block's transformed operand -> LDS (thread: one 16-byte channel group `s`, rows row0 + RSTEP * i)
    const int s = tid % SPR, row0 = tid / SPR;
    const float* tab = sTab + s * 24;
    constexpr bool SEQROWS = COLMAJ && KCH == 256;      // (two row blocks of 256 channels in registers at once spill: one after the other)
    constexpr int NRZ = (COLMAJ && !SEQROWS) ? RBN : 1;
    uint4 rz[NRZ][ALD], ry[NRZ][ALD];
    auto issue_rows = [&](int m0, auto z_tag) {
      constexpr int Z = decltype(z_tag)::value;
#pragma unroll
      for (int i = 0; i < ALD; ++i) {
        const int m = m0 + row0 + RSTEP * i;
        const unsigned off = m < a.M ? (unsigned)((m * KCH + s * 8) * 2) : OOB;
        rz[Z][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, off, 0, 0));
        ry[Z][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX2, off, 0, 0));
      }
    };
    auto commit_rows = [&](int m0, auto z_tag, int slot) {      // registers of set Z -> row-block image `slot`
      constexpr int Z = decltype(z_tag)::value;
      char* sAz = sA + slot * (PWR_BM * RB);
#pragma unroll
      for (int i = 0; i < ALD; ++i) {
        const int row = row0 + RSTEP * i, m = m0 + row;
        uint4 v = affine2_vec<T>(rz[Z][i], ry[Z][i], tab, tab + 8, tab + 16);
        if (m >= a.M) v = make_uint4(0, 0, 0, 0);          // (rows past M load as 0, which the affine map turns into gam)
        u32x4_t sv; sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
        __builtin_amdgcn_raw_buffer_store_b128(sv, rsrcDy, m < a.M ? (unsigned)((m * KCH + s * 8) * 2) : OOB, 0, 0);      // (empty descriptor without dy_out)
        *reinterpret_cast<uint4*>(sAz + row * RB + ((s ^ pwr_swz<RB>(row)) << 4)) = v;
      }
    };
    uint4 w[WRES ? 2 : 1][KS][2];
    unsigned wvoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wvoff[j] = (unsigned)(((32 * wave + chan_of(j, fr)) * KCH + 8 * fq) * 2);
    auto issue_w = [&](int ct, auto c_tag) {
      constexpr int C = decltype(c_tag)::value;
      const int sw = ct * PWR_BN * KCH * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) w[C][ks][j] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcW, wvoff[j], sw + ks * 64, 0));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // set-up: the first row block (COLMAJ: every row block of the range) and the weights are requested before the tables are built
    int cur_rb = lo / col_tiles;
    int cct = 0, cr = 0;                    // (COLMAJ) the compute waves' position in the walk
    int nxt = COLMAJ ? walk(cct, cr) : lo;  // the next item
    issue_rows(cur_rb * PWR_BM, I0{});
    if constexpr (NRZ > 1) { if (nrb > 1) issue_rows((cur_rb + 1) * PWR_BM, I1{}); }
    if constexpr (NRZ > 2) { if (nrb > 2) issue_rows((cur_rb + 2) * PWR_BM, I2{}); }
    if constexpr (NRZ > 3) { if (nrb > 3) issue_rows((cur_rb + 3) * PWR_BM, I3{}); }
    if constexpr (WRES) { issue_w(0, I0{}); issue_w(1, I1{}); }
    else issue_w(nxt % col_tiles, I0{});
    if constexpr (CTN == 0) { for (int c = tid; c < 2 * a.Ncol; c += CT) sStat[c] = 0.f; }
    {   // mask table: entry b, dword q = all-ones halves for bits 2q, 2q + 1 of b
      uint4 m;
      unsigned* mw = reinterpret_cast<unsigned*>(&m);
#pragma unroll
      for (int q = 0; q < 4; ++q) mw[q] = (((unsigned)tid >> (2 * q)) & 1u ? 0xffffu : 0u) | (((unsigned)tid >> (2 * q + 1)) & 1u ? 0xffff0000u : 0u);
      *reinterpret_cast<uint4*>(sLut + tid * 16) = m;
    }
    if (a.in_scale) {
      for (int c = tid; c < KCH; c += CT) {
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = a.in_scale[c]; t[8] = a.in_shift[c]; t[16] = a.pro_gam[c];
      }
    } else {
      const BnTot b = bn_tot_copy(a.pro_tot);
      bn_tot_foreach<CT>(b.tot, b.R, KCH, [&](int c, double sa, double sb) {
        float al, be, ga;
        bn_bwd_consts(sa, sb, b.inv_count, b.gamma[c], b.mean[c], b.invstd[c], al, be, ga);
        float* t = sTab + (c >> 3) * 24 + (c & 7);
        t[0] = al; t[8] = be; t[16] = ga;
      });
    }
    __syncthreads();
    commit_rows(cur_rb * PWR_BM, I0{}, 0);
    if constexpr (NRZ > 1) { if (nrb > 1) commit_rows((cur_rb + 1) * PWR_BM, I1{}, 1); }
    if constexpr (NRZ > 2) { if (nrb > 2) commit_rows((cur_rb + 2) * PWR_BM, I2{}, 2); }
    if constexpr (NRZ > 3) { if (nrb > 3) commit_rows((cur_rb + 3) * PWR_BM, I3{}, 3); }
    if constexpr (SEQROWS) {
      for (int r = 1; r < nrb; ++r) { issue_rows((cur_rb + r) * PWR_BM, I0{}); commit_rows((cur_rb + r) * PWR_BM, I0{}, r); }
    }
    __syncthreads();
    FRX_STAMP(1);
    bool rows_pending = false;
    float csum[NCT][8], csq[NCT][8];        // CTN > 0: the block's running sums per column tile; CTN == 0: one item's
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) { csum[c][e] = 0.f; csq[c][e] = 0.f; }

    // c_tag: the item's column tile where it indexes registers (CTN > 0), else 0
    // the lane reduction of one column tile's sums into the block's table (CTN == 0)
    auto flush_stats = [&](int ct) {
      lane16_butterfly<8, 8>(csum[0], csq[0], fr);
      if (fr < 8) {                       // lane (fq, fr < 8) owns column 32 wave + 8 fq + fr of every column tile: no other lane of the block adds to it
        const int col = ct * PWR_BN + 32 * wave + 8 * fq + fr;
        sStat[col] += csum[0][0];
        sStat[a.Ncol + col] += csq[0][0];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { csum[0][e] = 0.f; csq[0][e] = 0.f; }
    };
    // seq: the item's position in the walk; nct: the column tile of the NEXT item (-1: none)
    auto do_item = [&](int it, int seq, int rb, int ct, int nct, auto c_tag) {
      constexpr int C = decltype(c_tag)::value;
      constexpr int WC = WRES ? C : 0;
      const int m0 = rb * PWR_BM, n0 = ct * PWR_BN;
      const char* sAr = sA + (COLMAJ ? (rb - rb_lo) * (PWR_BM * RB) : 0);
      const char* sEs = sE + (seq % RING) * PWR_EBYTES;
      const char* sEi = sEs + (wave * 1024 + lane * 16);
      const char* sBi = sEs + 32768 + (wave * 256 + lane * 4);
      pwr_barrier();                      // item `it`'s operands are in its slot; every compute wave is done with item it - 1
      if constexpr (!COLMAJ) {
        if (rb != cur_rb) {               // (block-uniform) the next row block's operand replaces this one
          if (!rows_pending) issue_rows(m0, I0{});
          commit_rows(m0, I0{}, 0);
          pwr_barrier();
          cur_rb = rb; rows_pending = false;
        }
      }
      if constexpr (APF) {                // the row block after this one starts with the next item: request its rows now
        if (ct == col_tiles - 1 && it + 1 < hi) { issue_rows(m0 + PWR_BM, I0{}); rows_pending = true; }
      }
      // the item's epilogue operands leave LDS BEFORE the MFMA loop: a lone wave per SIMD has nothing else to cover the ~100
      // clocks of each LDS read with (12 reads per item ahead of dependent arithmetic were ~40 % of the epilogue's time)
      uint4 pea[ADD ? 4 : 1], pey[4];
      unsigned pmb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ADD) pea[i] = *reinterpret_cast<const uint4*>(sEi + (0 * 4 + i) * 4096);
        pey[i] = *reinterpret_cast<const uint4*>(sEi + (1 * 4 + i) * 4096);
        pmb[i] = *reinterpret_cast<const unsigned*>(sBi + i * 1024);
      }
      f32x4 acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 16 * i + fr;
          fa[i] = *reinterpret_cast<const uint4*>(sAr + row * RB + (((ks * 4 + fq) ^ pwr_swz<RB>(row)) << 4));
        }
        // operands swapped (weights first): D[row = channel][col = pixel] -- a lane ends up with 8 consecutive channels of a pixel
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w[WC][ks][j]), *reinterpret_cast<bf16x8*>(&fa[i]), acc[i][j], 0, 0, 0);
      }
      if constexpr (!WRES) {              // the next item's weight fragments (COLMAJ: only where the column tile changes)
        if (nct >= 0 && (!COLMAJ || nct != ct)) issue_w(nct, I0{});
      }
      uint4 pmk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pmk[i] = *reinterpret_cast<const uint4*>(sLut + ((pmb[i] >> (8 * fq)) & 255u) * 16);

      unsigned yo[4], ao[4];
      row_offsets(m0, n0, yo, ao);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8], yv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][e >> 2][e & 3];
        if constexpr (ADD) {
          const unsigned* q = reinterpret_cast<const unsigned*>(&pea[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(q[e] << 16); v[2 * e + 1] += __uint_as_float(q[e] & 0xffff0000u); }
        }
        {
          const unsigned* q = reinterpret_cast<const unsigned*>(&pey[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { yv[2 * e] = __uint_as_float(q[e] << 16); yv[2 * e + 1] = __uint_as_float(q[e] & 0xffff0000u); }
        }
        const uint4 mk = pmk[i];
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
        u32x4_t tw = *reinterpret_cast<u32x4_t*>(&t);
        tw[0] &= mk.x; tw[1] &= mk.y; tw[2] &= mk.z; tw[3] &= mk.w;      // the mask on the packed result: +0.0 where the merge output was <= 0
        __builtin_amdgcn_raw_buffer_store_b128(tw, rsrcY, yo[i], 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                            // statistics of what the next kernel reads
          const float vl = __uint_as_float(tw[e] << 16), vh = __uint_as_float(tw[e] & 0xffff0000u);
          csum[C][2 * e] += vl; csum[C][2 * e + 1] += vh;
          csq[C][2 * e] += vl * yv[2 * e]; csq[C][2 * e + 1] += vh * yv[2 * e + 1];
        }
      }
      if constexpr (CTN == 0) { if (nct != ct) flush_stats(ct); }      // (row-major walk: after every item; COLMAJ: once per column tile)
    };
    for (int seq = 0; nxt >= 0; ++seq) {
      const int it = nxt;
      nxt = COLMAJ ? walk(cct, cr) : (it + 1 < hi ? it + 1 : -1);
      const int rb = it / col_tiles, ct = it - rb * col_tiles;
      const int nct = nxt >= 0 ? nxt % col_tiles : -1;
      if constexpr (CTN == 0) do_item(it, seq, rb, ct, nct, I0{});
      else { if (ct == 0) do_item(it, seq, rb, ct, nct, I0{}); else do_item(it, seq, rb, ct, nct, I1{}); }
    }
    FRX_STAMP(2);
    if constexpr (CTN > 0) {              // the block's sums: one lane-reduction per column tile, then straight into the replicated totals
      const int rep = blockIdx.x & (a.stat_R - 1);
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        lane16_butterfly<8, 8>(csum[c], csq[c], fr);
        if (fr < 8) {
          const int col = c * PWR_BN + 32 * wave + 8 * fq + fr;
          const float s1 = csum[c][0], s2 = csq[c][0];
          if (s1 != 0.f || s2 != 0.f) {
            __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 0) * a.Ncol + col, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 1) * a.Ncol + col, a.e_invstd[col] * (s2 - a.e_mean[col] * s1), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
  if constexpr (CTN == 0) {
    __syncthreads();
    const int rep = blockIdx.x & (a.stat_R - 1);
    for (int c = tid; c < a.Ncol; c += PWR_NT) {
      const float s1 = sStat[c], s2 = sStat[a.Ncol + c];
      if (s1 != 0.f || s2 != 0.f) {
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 0) * a.Ncol + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 1) * a.Ncol + c, a.e_invstd[c] * (s2 - a.e_mean[c] * s1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef FRX_DBG_TIMES
  __builtin_amdgcn_s_waitcnt(0);
  FRX_STAMP(3);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The conv3-type FORWARD of a bottleneck block (1x1, middle width -> 4x: torchvision Bottleneck.conv3 behind bn2 + ReLU,
// backbones.py:16-18), same organisation without loaders -- its epilogue reads nothing: a row block's normalised input
// relu(scale * x + shift) is built once and stays in LDS, the block walks its (row block, column tile) items, the weights come
// from L2 straight into fragment registers (resident when the launch has two column tiles), the output is stored as bf16 and
// the BatchNorm statistics (sum, sum of squares of the rounded output) go to the replicated totals -- per column tile in
// registers for the whole block where the launch has two tiles, else through the per-item lane reduction and an LDS table.
// Four waves per block, several blocks per CU (the LDS footprint is the row block).
template <int KCH, int CTN>
__global__ __launch_bounds__(256, KCH <= 128 ? 3 : 2) void k_pw_rows_fwd(ConvArgs a, int items, int col_tiles) {
  typedef bf16_t T;
  constexpr int RB = KCH * 2, SPR = KCH / 8, KS = KCH / 32;
  constexpr int CT = 256;
  constexpr int ALD = KS, RSTEP = CT / SPR;
  constexpr bool APF = KCH <= 128;
  constexpr bool WRES = CTN == 2;
  constexpr int NCT = CTN > 0 ? CTN : 1;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(KCH == 64 || KCH == 128 || KCH == 256, "middle widths served");
  static_assert(CTN == 0 || CTN == 2, "column tiles held in registers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  float* sTab = reinterpret_cast<float*>(sA + PWR_BM * RB);       // [KCH / 8][scale, shift][8]
  float* sStat = sTab + 2 * KCH;                                  // CTN == 0: [2][Ncol]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int lo = (int)((long)items * blockIdx.x / gridDim.x), hi = (int)((long)items * (blockIdx.x + 1) / gridDim.x);
  if (lo >= hi) return;
  FRX_STAMP(0);
  const unsigned ybytes = (unsigned)a.M * (unsigned)a.Ncol * 2u;
  const unsigned rstep = 16u * (unsigned)a.Ncol * 2u;
  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, a.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ybytes, 0x00020000);
  const int s = tid % SPR, row0 = tid / SPR;
  const float* tab = sTab + s * 16;
  uint4 rx[ALD];
  auto issue_rows = [&](int m0) {
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      const int m = m0 + row0 + RSTEP * i;
      rx[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcX, m < a.M ? (unsigned)((m * KCH + s * 8) * 2) : OOB, 0, 0));
    }
  };
  auto commit_rows = [&](int m0) {
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
      const int row = row0 + RSTEP * i, m = m0 + row;
      uint4 v = bn_relu_vec<T>(rx[i], tab, tab + 8, a.in_relu);
      if (m >= a.M) v = make_uint4(0, 0, 0, 0);          // (rows past M load as 0, which the prologue turns into f(0))
      *reinterpret_cast<uint4*>(sA + row * RB + ((s ^ pwr_swz<RB>(row)) << 4)) = v;
    }
  };
  uint4 w[WRES ? 2 : 1][KS][2];
  unsigned wvoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wvoff[j] = (unsigned)(((32 * wave + chan_of(j, fr)) * KCH + 8 * fq) * 2);
  auto issue_w = [&](int ct, auto c_tag) {
    constexpr int C = decltype(c_tag)::value;
    const int sw = ct * PWR_BN * KCH * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) w[C][ks][j] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rsrcW, wvoff[j], sw + ks * 64, 0));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  int cur_rb = lo / col_tiles;
  issue_rows(cur_rb * PWR_BM);
  if constexpr (WRES) { issue_w(0, I0{}); issue_w(1, I1{}); }
  else issue_w(lo % col_tiles, I0{});
  if constexpr (CTN == 0) { for (int c = tid; c < 2 * a.Ncol; c += CT) sStat[c] = 0.f; }
  if (a.in_scale) {
    for (int c = tid; c < KCH; c += CT) {
      float* t = sTab + (c >> 3) * 16 + (c & 7);
      t[0] = a.in_scale[c]; t[8] = a.in_shift[c];
    }
  } else {                                // the producer's replicated totals -> scale / shift (bn_tot.h)
    const BnTot b = bn_tot_copy(a.in_tot);
    bn_tot_foreach<CT>(b.tot, b.R, KCH, [&](int c, double sm, double sq) {
      float mean, invstd, sc, sh; double var;
      bn_fwd_consts(sm, sq, b.inv_count, b.gamma[c], b.beta[c], b.eps, mean, invstd, sc, sh, var);
      float* t = sTab + (c >> 3) * 16 + (c & 7);
      t[0] = sc; t[8] = sh;
    });
  }
  __syncthreads();
  commit_rows(cur_rb * PWR_BM);
  pwr_barrier();
  FRX_STAMP(1);
  bool rows_pending = false;
  float csum[NCT][8], csq[NCT][8];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[c][e] = 0.f; csq[c][e] = 0.f; }

  auto do_item = [&](int it, int rb, int ct, auto c_tag) {
    constexpr int C = decltype(c_tag)::value;
    constexpr int WC = WRES ? C : 0;
    const int m0 = rb * PWR_BM, n0 = ct * PWR_BN;
    if (rb != cur_rb) {                 // (block-uniform) the next row block's operand replaces this one
      if (!rows_pending) issue_rows(m0);
      pwr_barrier();                    // every wave is done reading the old one
      commit_rows(m0);
      pwr_barrier();
      cur_rb = rb; rows_pending = false;
    }
    if constexpr (APF) {
      if (ct == col_tiles - 1 && it + 1 < hi) { issue_rows(m0 + PWR_BM); rows_pending = true; }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * i + fr;
        fa[i] = *reinterpret_cast<const uint4*>(sA + row * RB + (((ks * 4 + fq) ^ pwr_swz<RB>(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w[WC][ks][j]), *reinterpret_cast<bf16x8*>(&fa[i]), acc[i][j], 0, 0, 0);
    }
    if constexpr (!WRES) { if (it + 1 < hi) issue_w((it + 1) % col_tiles, I0{}); }
    const unsigned y0 = (unsigned)(((m0 + fr) * a.Ncol + n0 + 32 * wave + 8 * fq) * 2);
    if constexpr (CTN == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { csum[0][e] = 0.f; csq[0][e] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (bf16_t)acc[i][e >> 2][e & 3];
      u32x4_t tw = *reinterpret_cast<u32x4_t*>(&t);
      __builtin_amdgcn_raw_buffer_store_b128(tw, rsrcY, m0 + 16 * i + fr < a.M ? y0 + (unsigned)i * rstep : OOB, 0, 0);
      if (m0 + 16 * i + fr >= a.M) { tw[0] = 0u; tw[1] = 0u; tw[2] = 0u; tw[3] = 0u; }      // (rows past M are exactly 0 anyway: zero operand rows)
#pragma unroll
      for (int e = 0; e < 4; ++e) {                              // statistics of what the next kernel reads
        const float vl = __uint_as_float(tw[e] << 16), vh = __uint_as_float(tw[e] & 0xffff0000u);
        csum[C][2 * e] += vl; csum[C][2 * e + 1] += vh;
        csq[C][2 * e] += vl * vl; csq[C][2 * e + 1] += vh * vh;
      }
    }
    if constexpr (CTN == 0) {
      lane16_butterfly<8, 8>(csum[0], csq[0], fr);
      if (fr < 8) {
        const int col = n0 + 32 * wave + 8 * fq + fr;
        sStat[col] += csum[0][0];
        sStat[a.Ncol + col] += csq[0][0];
      }
    }
  };
  for (int it = lo; it < hi; ++it) {
    const int rb = it / col_tiles, ct = it - rb * col_tiles;
    if constexpr (CTN == 0) do_item(it, rb, ct, I0{});
    else { if (ct == 0) do_item(it, rb, ct, I0{}); else do_item(it, rb, ct, I1{}); }
  }
  FRX_STAMP(2);
  const int rep = blockIdx.x & (a.stat_R - 1);
  if constexpr (CTN > 0) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      lane16_butterfly<8, 8>(csum[c], csq[c], fr);
      if (fr < 8) {
        const int col = c * PWR_BN + 32 * wave + 8 * fq + fr;
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 0) * a.Ncol + col, csum[c][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 1) * a.Ncol + col, csq[c][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    __syncthreads();
    for (int c = tid; c < a.Ncol; c += CT) {
      const float s1 = sStat[c], s2 = sStat[a.Ncol + c];
      if (s1 != 0.f || s2 != 0.f) {
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 0) * a.Ncol + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.stat_tot + ((long)rep * 2 + 1) * a.Ncol + c, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef FRX_DBG_TIMES
  __builtin_amdgcn_s_waitcnt(0);
  FRX_STAMP(3);
#endif
}

bool pw_rows_fwd_ok(const ConvArgs& a, int dtype, int epi) {
  if (const char* e = getenv("FRX_PW_ROWS")) { if (!(atoi(e) & 2)) return false; }
  const bool pw = a.mode == MODE_FWD && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;
  return dtype == FRX_BF16 && pw && (a.in_scale || a.in_tot.tot) && epi == EPI_STATS && a.stat_tot && !a.stat_partial && !a.out_f32 && !a.bias &&
         !a.dy_out && !a.addend && (a.Kc == 64 || a.Kc == 128 || (a.Kc == 256 && getenv("FRX_PWR_FWD256"))) && a.Ncol % PWR_BN == 0 && a.Ncol >= 2 * a.Kc;
}

template <int KCH, int CTN>
static void launch_fwd_one(hipStream_t st, const ConvArgs& a, int items, int col_tiles, unsigned lds) {
  static bool attr_done[64] = {false};      // (per device: the attribute belongs to the function ON a device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw_rows_fwd<KCH, CTN>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_done[dev & 63] = true;
  }
  const int cap = 256 * (KCH <= 128 ? 3 : 2), grid = items < cap ? items : cap;
  hipLaunchKernelGGL((k_pw_rows_fwd<KCH, CTN>), dim3(grid), dim3(256), lds, st, a, items, col_tiles);
}

int launch_pw_rows_fwd(hipStream_t st, const ConvArgs& a) {
  const int col_tiles = a.Ncol / PWR_BN, items = cdiv(a.M, PWR_BM) * col_tiles;
  const unsigned lds = (unsigned)(PWR_BM * a.Kc * 2 + 2 * a.Kc * 4 + 2 * a.Ncol * 4);
  int ctn = (a.Kc == 64 && col_tiles == 2) ? 2 : 0;
  if (const char* e = getenv("FRX_PWR_CTN")) { if (atoi(e) == 0) ctn = 0; }      // (tuning aid, read per launch)
  note_igemm_launch(PWR_BM, PWR_BN, 4, 64, 0, MODE_FWD, 1, EPI_STATS, 0, 1, 2);
  if (a.Kc == 64) { if (ctn == 2) launch_fwd_one<64, 2>(st, a, items, col_tiles, lds); else launch_fwd_one<64, 0>(st, a, items, col_tiles, lds); }
  else if (a.Kc == 128) launch_fwd_one<128, 0>(st, a, items, col_tiles, lds);
  else launch_fwd_one<256, 0>(st, a, items, col_tiles, lds);
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

bool pw_rows_dgrad_ok(const ConvArgs& a, int dtype, int epi) {
  if (const char* e = getenv("FRX_PW_ROWS")) { if (!(atoi(e) & 1)) return false; }      // (bit 0: input gradients, bit 1: forwards)
  const bool pw = a.mode == MODE_DGRAD && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && !a.s2c;
  return dtype == FRX_BF16 && pw && a.X2 && epi == EPI_BNBWD_OUT && a.e_bits && a.stat_tot && !a.stat_partial && !a.out_f32 &&
         (a.Kc == 64 || a.Kc == 128 || a.Kc == 256) && a.Ncol % PWR_BN == 0 && a.Ncol >= 2 * a.Kc &&
         (!a.addend || a.add_stride == 2 || a.add_stride == 0 || a.add_stride == 1);
}

template <int KCH, bool ADD, int RING, int CTN, int RBN>
static void launch_one(hipStream_t st, const ConvArgs& a, int items, int col_tiles, unsigned lds) {
  static bool attr_done[64] = {false};      // (per device: the attribute belongs to the function ON a device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_done[dev & 63]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw_rows_dgrad<KCH, ADD, RING, CTN, RBN>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    attr_done[dev & 63] = true;
  }
  const int grid = items < 256 ? items : 256;
  hipLaunchKernelGGL((k_pw_rows_dgrad<KCH, ADD, RING, CTN, RBN>), dim3(grid), dim3(PWR_NT), lds, st, a, items, col_tiles);
}

int launch_pw_rows_dgrad(hipStream_t st, const ConvArgs& a) {
  const int col_tiles = a.Ncol / PWR_BN, items = cdiv(a.M, PWR_BM) * col_tiles;
  const bool add = a.addend != nullptr;
  // statistics and weights of both column tiles in registers: layer1's shape
  int ctn = ((a.Kc == 64 || a.Kc == 128) && col_tiles == 2) ? 2 : 0;      // (four tiles cost more registers than two waves per SIMD have)
  if (const char* e = getenv("FRX_PWR_CTN")) { if (atoi(e) == 0) ctn = 0; }      // (tuning aid, read per launch)
  // column-tile-major walk over resident row blocks where a block's range spans few of them: layer2's shape (128 channels,
  // four row blocks of 16 KB; stand-alone 48.2 -> 44.4 us).  (256 channels, two row blocks: 38.3 -> 41.5 us -- the second
  // row block's build at set-up costs more than six items save; not instantiated.)
  int rbn = a.Kc == 128 ? 4 : 0;
  const int per_block = cdiv(items, items < 256 ? items : 256);
  if (ctn || cdiv(col_tiles - 1 + per_block, col_tiles) > rbn) rbn = 0;
  if (const char* e = getenv("FRX_PWR_COLMAJ")) { if (atoi(e) == 0) rbn = 0; }   // (tuning aid, read per launch)
  const int ring = rbn ? 2 : 3;       // (2, 3 and 4 slots measured alike: the compute waves are the bound)
  const unsigned lds = pw_rows_lds(a.Kc, a.Ncol, ring, rbn);
  FRX_CHECK_ARG(lds <= 159u * 1024u, "pw_rows: %u bytes of LDS", lds);
  note_igemm_launch(PWR_BM, PWR_BN, 8, 64, ring, MODE_DGRAD, 2, EPI_BNBWD_OUT, add, 1, 2);
#define FRX_PWR4(K_, R_, C_, B_) do { if (add) launch_one<K_, true, R_, C_, B_>(st, a, items, col_tiles, lds); else launch_one<K_, false, R_, C_, B_>(st, a, items, col_tiles, lds); } while (0)
  if (a.Kc == 64) { if (ctn == 2) FRX_PWR4(64, 3, 2, 0); else FRX_PWR4(64, 3, 0, 0); }
  else if (a.Kc == 128) { if (ctn == 2) FRX_PWR4(128, 3, 2, 0); else if (rbn) FRX_PWR4(128, 2, 0, 4); else FRX_PWR4(128, 3, 0, 0); }
  else FRX_PWR4(256, 3, 0, 0);
#undef FRX_PWR4
  FRX_LAUNCH_CHECK();
  return FRX_OK;
}

}  // namespace frx

FRX_DBG_EXPORT(frx_debug_times_pw_rows)
